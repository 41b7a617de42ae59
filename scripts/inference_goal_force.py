#!/usr/bin/env python3
"""Goal-Force inference driver on the MI355X pipeline — mirror of the reference's launchers
scripts/inference/inference_goal_force.py (INF) and scripts/inference/inference_canny_edge_control.py (INFC): the SAME command-line
arguments (INF:38-56: --device_id --world_size --seed --control_signal_type {goal_force,canny_edge} --model_ckpt_path --example_paths),
constants (INF:27-32; NUM_FRAMES 81 / 49 by control signal, INFC:30), CSV sharding (INF:41-44, 114; scripts/inference/utils.py:25-57),
dataset overrides (INF:137-146), output directory `<ckpt dir>/step-<N>-videos` and file-name roots (INF:73-76, 178-186; INFC:160-163),
and the pipe(...) call (INF:206-215, INFC:177-186).

    python scripts/inference_goal_force.py --device_id 0 --world_size 8 --seed 0 --control_signal_type goal_force \
        --model_ckpt_path checkpoints/.../step-N.safetensors --example_paths a.csv b.csv

The reference hard-codes ./models/Wan-AI/... for the experts, the umT5 encoder, the VAE and the tokenizer (INF:81-106); here those
are the DEFAULTS of --dit_high / --dit_low / --text_encoder / --vae / --tokenizer, so the reference's shell scripts run unchanged.
Additions (not in the reference): --synthetic (random weights of the real shapes and seeded prompt embeddings: no checkpoints exist in
the build container), --layers, --num_inference_steps, --output_dir.  Frames are written as PNGs into `<name>/` where the reference
writes `<name>.mp4` (mp4 writing is host I/O through imageio, which this image lacks).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# dmabuf IPC for RCCL on this driver (goal_force_amd/distributed.py::ensure_ipc_env): set before anything touches the GPU
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

CONTROLNET_NUM_LAYERS = 10   # INF:27-32
NUM_FRAMES = 81
NUM_FRAMES_CANNY = 49     # INFC:30
NEGATIVE_PROMPT = ("色调艳丽，过曝，静态，细节模糊不清，字幕，风格，作品，画作，画面，静止，整体发灰，最差质量，低质量，JPEG压缩残留，丑陋的，残缺的，"
                   "多余的手指，画得不好的手部，画得不好的脸部，畸形的，毁容的，形态畸形的肢体，手指融合，静止不动的画面，杂乱的背景，三条腿，背景人很多，倒着走")


MODELS = "./models/Wan-AI"
DEFAULTS = dict(                                                            # INF:84-103
    dit_high=[f"{MODELS}/Wan2.2-I2V-A14B/high_noise_model/diffusion_pytorch_model-0000{i}-of-00006.safetensors" for i in range(1, 7)],
    dit_low=[f"{MODELS}/Wan2.2-I2V-A14B/low_noise_model/diffusion_pytorch_model-0000{i}-of-00006.safetensors" for i in range(1, 7)],
    text_encoder=f"{MODELS}/Wan2.1-T2V-1.3B/models_t5_umt5-xxl-enc-bf16.pth", vae=f"{MODELS}/Wan2.1-T2V-1.3B/Wan2.1_VAE.pth",
    tokenizer=f"{MODELS}/Wan2.1-T2V-1.3B/google/umt5-xxl")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--device_id", type=int, default=0, help="Device ID")                                        # INF:41
    ap.add_argument("--world_size", type=int, default=1, help="Total number of devices/processes (for CSV partitioning)")
    ap.add_argument("--seed", type=int, default=0, help="Seed for video generation (default: 0)")
    ap.add_argument("--control_signal_type", choices=["goal_force", "canny_edge"], default="goal_force",        # INF:47-50
                    help="Type of control signal to use (default: goal_force)")
    ap.add_argument("--model_ckpt_path", "--controlnet_checkpoint", dest="model_ckpt_path", default=None,       # INF:51 (required there)
                    help="Path to the ControlNet checkpoint (step-N.safetensors); required unless --synthetic")
    ap.add_argument("--example_paths", nargs="+", required=True,
                    help="CSV file(s): goal_force rows (README.md:92-107) or canny_edge rows (image, control_video, caption)")
    ap.add_argument("--output_dir", default=None, help="default: <dir of --model_ckpt_path>/step-<N>-videos (INF:73-76); ./outputs with --synthetic")
    ap.add_argument("--num_inference_steps", type=int, default=50)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--layers", type=int, default=40, help=argparse.SUPPRESS)
    for n in ("dit_high", "dit_low"):
        ap.add_argument("--" + n, nargs="*", default=DEFAULTS[n])
    for n in ("vae", "text_encoder", "tokenizer"):
        ap.add_argument("--" + n, default=DEFAULTS[n])
    a = ap.parse_args(argv)
    if a.model_ckpt_path is None and not a.synthetic:
        ap.error("the following arguments are required: --model_ckpt_path")
    return a


def output_location(a):
    """(directory, step tag) as INF:73-76: <ckpt dir>/step-<N>-videos with N read from `...-<N>.safetensors`."""
    if a.model_ckpt_path is None:
        return a.output_dir or "outputs", "synthetic"
    step_num = os.path.basename(a.model_ckpt_path).split(".safetensors")[0].split("-")[-1]
    return a.output_dir or os.path.join(os.path.dirname(a.model_ckpt_path), f"step-{step_num}-videos"), step_num


def goal_force_name(step_num, data, seed):
    """The file-name root of one goal-force sample (INF:178-186)."""
    s = f"step-{step_num}_{data['file_id']}"
    s += f"__prj_coords_{data['x_pos']:.2f}_{data['y_pos']:.2f}"
    s += f"__tgt_coords_{data['target_x_pos']:.2f}_{data['target_y_pos']:.2f}"
    s += f"__prj_mass_{data['masses']['projectile']:.1f}"
    s += f"__tgt_mass_{data['masses']['target']:.1f}"
    s += f"__prj_force_{data['force']:.1f}__prj_angle_{data['angle']:.1f}"
    s += f"__tgt_indirect_force_{data['target_indirect_force']:.1f}__tgt_indirect_angle_{data['target_indirect_angle']:.1f}"
    return s + f"__seed_{seed}"


def main(argv=None):
    a = parse_args(argv)
    num_frames = NUM_FRAMES if a.control_signal_type == "goal_force" else NUM_FRAMES_CANNY

    import json
    import numpy as np
    import torch
    from PIL import Image
    from goal_force_amd.canny import ControlSignalDataset_CannyEdge
    from goal_force_amd.dit import A14B_CONFIG
    from goal_force_amd.distributed import split_list_across_devices_contiguous
    from goal_force_amd.force_map import ControlSignalDataset_Balls
    from goal_force_amd.pipeline import ModelConfig, WanVideoPipeline, build_random_controlnet, build_random_expert
    from goal_force_amd.vae import WanVideoVAE

    torch.set_grad_enabled(False)
    if a.world_size > 1 and "OMP_NUM_THREADS" not in os.environ:
        # one process per device, started side by side by the shell script: without a budget each would run its host work (checkpoint
        # casts, VAE / tokenizer set-up) on ALL cores and the processes thrash each other (measured: 70 s instead of 0.5 s at 8 ranks)
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // a.world_size)))
    # INF:62-67: with HIP_/CUDA_VISIBLE_DEVICES set the isolated GPU appears as cuda:0, otherwise this process takes
    # cuda:{device_id}.  The C-ABI launches go to the CURRENT HIP device, so it is selected before anything is built.
    isolated = any(v in os.environ for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
    dev = torch.device("cuda", 0 if isolated else a.device_id)
    if dev.index >= torch.cuda.device_count():
        raise SystemExit(f"--device_id {a.device_id}: only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(dev)
    print(f"[Device {a.device_id}] world size {a.world_size}, device {dev}, seed {a.seed}, control signal {a.control_signal_type}")
    step_dir, step_num = output_location(a)
    os.makedirs(step_dir, exist_ok=True)
    if a.synthetic:
        cfg = dict(A14B_CONFIG, num_layers=a.layers)
        n_cn = min(CONTROLNET_NUM_LAYERS, a.layers)
        pipe = WanVideoPipeline.from_modules(build_random_expert(cfg, 100, dev), build_random_expert(cfg, 200, dev),
                                             build_random_controlnet(n_cn, cfg, 300, dev),
                                             build_random_controlnet(n_cn, cfg, 400, dev, zero_convs_zero=True),
                                             vae=WanVideoVAE().to(torch.bfloat16).to(dev), device=dev)
    else:
        # the reference's call (INF:81-106): both experts as shard lists, the umT5 encoder and the VAE .pth, the tokenizer
        pipe = WanVideoPipeline.from_pretrained(
            torch_dtype=torch.bfloat16, device=dev,
            tokenizer_config=ModelConfig(model_id="Wan-AI/Wan2.1-T2V-1.3B", origin_file_pattern="google/*", path=a.tokenizer),
            model_configs=[ModelConfig(path=a.dit_high), ModelConfig(path=a.dit_low), ModelConfig(path=a.text_encoder),
                           ModelConfig(path=a.vae)],
            controlnet=True, controlnet_num_layers=CONTROLNET_NUM_LAYERS)
        pipe.load_controlnet_weights(pipe.controlnet, a.model_ckpt_path, torch_dtype=torch.bfloat16)   # INF:108
    pipe.enable_vram_management()   # INF:111 (accepted no-op: everything is resident)

    def synthetic_context():
        extra = {}
        if a.synthetic:
            g = torch.Generator().manual_seed(a.seed)
            for k in ("context_posi", "context_nega"):
                c = torch.randn((1, 512, 4096), generator=g)
                c[:, 40:] = 0
                extra[k] = c.to(torch.bfloat16).to(dev)
        return extra

    def save_frames(video, root):
        out = os.path.join(step_dir, root)
        os.makedirs(out, exist_ok=True)
        for t, frame in enumerate(video):
            frame.save(os.path.join(out, f"{t:03d}.png"))
        print(f"[device {a.device_id}] wrote {len(video)} frames to {out}")

    device_examples = split_list_across_devices_contiguous(a.example_paths, a.world_size, a.device_id)
    print(f"[Device {a.device_id}, seed {a.seed}] processing {len(device_examples)} of {len(a.example_paths)} examples: {device_examples}")
    for csv in device_examples:
        base_path = os.path.dirname(csv)
        if a.control_signal_type == "canny_edge":
            # INFC:121-186: rows (image, control_video, caption); the control clip is a pre-computed Canny video under canny-videos/,
            # loaded by the dataset's own video operator (centre crop + resize to 480 x 832, 49 frames), x / 127.5 - 1 in bf16
            import pandas
            op = ControlSignalDataset_CannyEdge.default_video_operator(
                base_path=os.path.join(base_path, "canny-videos"), max_pixels=921600, height=480, width=832, height_division_factor=16,
                width_division_factor=16, num_frames=num_frames, time_division_factor=4, time_division_remainder=1)
            for _, row in pandas.read_csv(csv).iterrows():
                image_path = os.path.join(base_path, "images", row["image"])
                if not os.path.exists(image_path):
                    raise FileNotFoundError(f"Image file not found: {image_path}")
                input_image = Image.open(image_path).convert("RGB")
                frames = op(row["control_video"])
                if frames is None:
                    raise SystemExit(f"control video {row['control_video']} could not be read")
                control = (torch.from_numpy(np.array(frames)).to(torch.float32) / 127.5 - 1.0).to(torch.bfloat16)     # INFC:154-156
                root = str(row["control_video"]).split("_canny.mp4")[0].split(".mp4")[0]                                # INFC:159
                input_image.save(os.path.join(step_dir, f"{root}-image-condition.png"))
                video = pipe(prompt=row["caption"], negative_prompt=NEGATIVE_PROMPT, input_image=input_image, num_frames=num_frames,
                             seed=a.seed, tiled=True, controlnet=True, control_signal_video=control.to(dev),
                             num_inference_steps=a.num_inference_steps, **synthetic_context())
                save_frames(video, f"{root}-canny-output")
            continue
        ds = ControlSignalDataset_Balls(base_path=base_path, metadata_path=csv, is_validation_dataset=True,
                                        num_frames=num_frames, height=480, width=832, device=dev)
        ds.min_mass, ds.max_mass, ds.min_force, ds.max_force = 1.0, 4.0, 30.0, 400.0        # INF:137-142
        ds.min_indirect_force, ds.max_indirect_force = ds.min_force, ds.max_force          # INF:145-146
        for i in range(len(ds)):
            data = ds[i]
            assert len(data["video"]) == 1                                                  # INF:155
            root = goal_force_name(step_num, data, a.seed)
            data["video"][0].save(os.path.join(step_dir, f"{root}-image_condition.png"))    # INF:196
            with open(os.path.join(step_dir, f"{root}-text.json"), "w") as f:               # INF:203-205
                json.dump({"text_prompt": data["prompt"]}, f, indent=4)
            video = pipe(prompt=data["prompt"], negative_prompt=NEGATIVE_PROMPT, input_image=data["video"][0].convert("RGB"),
                         num_frames=num_frames, seed=a.seed, tiled=True, controlnet=True,
                         control_signal_video=data["control_video"], num_inference_steps=a.num_inference_steps, **synthetic_context())
            save_frames(video, root)


if __name__ == "__main__":
    main()
