#!/usr/bin/env python3
"""ControlNet training launcher on the MI355X training step — mirror of the reference's scripts/train/train.py: the SAME command
line (`wan_parser`, src/goal_force/utils.py:854-900), dataset mux (train.py:126-197), training module (train.py:12-124: load the
models, freeze all but `--trainable_models controlnet`, scheduler in training mode, optional `--controlnet_checkpoint`) and loop
(`launch_training_task`, utils.py:734-826).  One process per GPU: where the reference starts `accelerate launch` with a DeepSpeed
ZeRO-2 config, start this with `python -m torch.distributed.run --nproc-per-node N scripts/train.py ...` (gradients are averaged over
RCCL, goal_force_amd/training.py::allreduce_gradients; 288 GB of HBM hold the frozen expert, the ControlNet, its fp32 moments and the
activations of an 81-frame item without ZeRO or offload).

    python scripts/train.py --dataset_base_path balls dominos plants --dataset_metadata_path balls.csv dominos.csv plants.csv \\
        --control_signal_type direct_force_and_goal_force_and_mass --controlnet_num_layers 10 --height 480 --width 832 --num_frames 81 \\
        --model_paths '["...high_noise_model shards...", "models_t5_umt5-xxl-enc-bf16.pth", "Wan2.1_VAE.pth"]' --learning_rate 1e-5 \\
        --num_epochs 2 --save_steps 500 --trainable_models controlnet --extra_inputs input_image --max_timestep_boundary 0.358 \\
        --max_grad_norm 1 --p_mask_out_masses 0.5 --p_mask_out_direct_force 0.5 --p_mask_out_indirect_force 0.5

Clips are decoded by cv2 as in the reference when it is importable; this image has neither cv2 nor imageio, so clips may also be
directories of frame images or `.npy` files (goal_force_amd/force_map.py::load_video_frames)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MODELS = "./models/Wan-AI"


def model_configs(args):
    """parse_model_configs / parse_model_configs_offline (utils.py:495-530): `--model_paths` is a JSON list of files or shard lists;
    `--model_id_with_origin_paths "id:pattern,..."` resolves under ./models/<id>/<pattern> (no downloads here)."""
    from goal_force_amd.pipeline import ModelConfig
    cfgs = []
    if args.model_paths is not None:
        cfgs += [ModelConfig(path=p) for p in json.loads(args.model_paths)]
    if args.model_id_with_origin_paths is not None:
        for item in args.model_id_with_origin_paths.split(","):
            mid, pattern = item.split(":")
            cfgs.append(ModelConfig(model_id=mid, origin_file_pattern=pattern))
    return cfgs


class WanTrainingModule:
    """train.py:12-124 around the HIP pipeline: what launch_training_task needs of it (`.pipe`, `.extra_inputs`, the timestep
    boundaries)."""

    def __init__(self, args, device):
        import torch
        from goal_force_amd.pipeline import ModelConfig, WanVideoPipeline
        for name in ("lora_base_model", "lora_checkpoint"):
            if getattr(args, name) is not None:
                raise NotImplementedError(f"--{name}: LoRA training is outside the Goal-Force path (the scripts train the ControlNet)")
        if args.apply_strided_controlnet or args.controlnet_stride is not None:
            raise NotImplementedError("strided ControlNet is not used by Goal Force")
        if args.control_signal_type not in ("canny_edge", "direct_force_and_goal_force_and_mass"):
            raise NotImplementedError(args.control_signal_type)                                            # train.py:33-38
        if (args.trainable_models or "") != "controlnet":
            raise NotImplementedError("--trainable_models controlnet is what the HIP training step differentiates")
        self.pipe = WanVideoPipeline.from_pretrained(
            torch_dtype=torch.bfloat16, device=device, model_configs=model_configs(args), controlnet=True,
            controlnet_num_layers=args.controlnet_num_layers,
            tokenizer_config=ModelConfig(model_id="Wan-AI/Wan2.1-T2V-1.3B", origin_file_pattern="google/*",
                                         path=f"{MODELS}/Wan2.1-T2V-1.3B/google/umt5-xxl"))               # train.py:43-55
        self.pipe.scheduler.set_timesteps(1000, training=True)                                             # utils.py:560
        self.pipe.freeze_except([] if args.trainable_models is None else args.trainable_models.split(","))   # utils.py:563
        if args.controlnet_checkpoint is not None:                                                         # utils.py:586-590
            self.pipe.load_controlnet_weights(self.pipe.controlnet, args.controlnet_checkpoint, torch_dtype=torch.bfloat16)
            print(f"ControlNet checkpoint loaded: {args.controlnet_checkpoint}, total {len(self.pipe.controlnet.state_dict())} keys")
        else:
            print("No ControlNet checkpoint provided. Starting training from scratch.")
        self.extra_inputs = args.extra_inputs.split(",") if args.extra_inputs is not None else []
        self.max_timestep_boundary, self.min_timestep_boundary = args.max_timestep_boundary, args.min_timestep_boundary


def main(argv=None):
    from goal_force_amd import training as tr
    args = tr.wan_parser().parse_args(argv)
    import torch
    from goal_force_amd.distributed import init_from_env
    rank, local, world = init_from_env()
    torch.cuda.set_device(local)
    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // world)))
    device = torch.device("cuda", local)
    dataset = tr.get_dataset(args, device=device)                          # (launch_training_task shards it over the ranks, as accelerator.prepare does)
    model = WanTrainingModule(args, device)
    log = (lambda rec, step: print(f"[step {step}] " + ", ".join(f"{k} {v:.6g}" for k, v in rec.items()), flush=True)) if rank == 0 else None
    tr.launch_training_task(dataset, model, args=args, log=log)


if __name__ == "__main__":
    main()
