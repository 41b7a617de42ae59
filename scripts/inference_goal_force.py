#!/usr/bin/env python3
"""Goal-Force inference driver on the MI355X pipeline — mirror of the reference's
scripts/inference/inference_goal_force.py (INF): same constants (INF:27-32), CSV sharding by
--device_id/--world_size (INF:41-44, 114; scripts/inference/utils.py:25-57), dataset overrides (INF:137-146) and
pipe(...) call (INF:206-215).

    python scripts/inference_goal_force.py --example_paths a.csv b.csv --device_id 0 --world_size 8 \
        --dit_high ckpt/high/*.safetensors --dit_low ckpt/low/*.safetensors --vae Wan2.1_VAE.pth \
        --text_encoder models_t5_umt5-xxl-enc-bf16.pth --tokenizer google/umt5-xxl --controlnet_checkpoint step-N.safetensors

With --synthetic the weights are random (no checkpoints exist in the build container) and the prompt embeddings
are seeded noise; everything else (force maps, VAE encode of the force map and first frame, 50-step loop, VAE decode)
runs for real.  Frames are written as PNGs (imageio/mp4 writing is host-side I/O outside the path).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# dmabuf IPC for RCCL on this driver (goal_force_amd/distributed.py::ensure_ipc_env): set before anything touches the GPU
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

CONTROLNET_NUM_LAYERS = 10   # INF:27-32
NUM_FRAMES = 81
NEGATIVE_PROMPT = ("色调艳丽，过曝，静态，细节模糊不清，字幕，风格，作品，画作，画面，静止，整体发灰，最差质量，低质量，JPEG压缩残留，丑陋的，残缺的，"
                   "多余的手指，画得不好的手部，画得不好的脸部，畸形的，毁容的，形态畸形的肢体，手指融合，静止不动的画面，杂乱的背景，三条腿，背景人很多，倒着走")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--example_paths", nargs="+", required=True)
    ap.add_argument("--device_id", type=int, default=0)
    ap.add_argument("--world_size", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--output_dir", default="outputs")
    ap.add_argument("--num_inference_steps", type=int, default=50)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--layers", type=int, default=40, help=argparse.SUPPRESS)
    for n in ("dit_high", "dit_low", "vae", "text_encoder", "tokenizer", "controlnet_checkpoint"):
        ap.add_argument("--" + n, nargs="*" if n.startswith("dit") else None, default=None)
    a = ap.parse_args()

    import torch
    from goal_force_amd.dit import A14B_CONFIG
    from goal_force_amd.distributed import split_list_across_devices_contiguous
    from goal_force_amd.force_map import ControlSignalDataset_Balls
    from goal_force_amd.pipeline import ModelConfig, WanVideoPipeline, build_random_controlnet, build_random_expert
    from goal_force_amd.vae import WanVideoVAE

    torch.set_grad_enabled(False)
    # INF:62-67: with HIP_/CUDA_VISIBLE_DEVICES set the isolated GPU appears as cuda:0, otherwise this process takes
    # cuda:{device_id}.  The C-ABI launches go to the CURRENT HIP device, so it is selected before anything is built.
    isolated = any(v in os.environ for v in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
    dev = torch.device("cuda", 0 if isolated else a.device_id)
    if dev.index >= torch.cuda.device_count():
        raise SystemExit(f"--device_id {a.device_id}: only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(dev)
    print(f"[Device {a.device_id}] world size {a.world_size}, device {dev}, seed {a.seed}")
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
        pipe.load_controlnet_weights(pipe.controlnet, a.controlnet_checkpoint, torch_dtype=torch.bfloat16)   # INF:108
    pipe.enable_vram_management()   # INF:111 (accepted no-op: everything is resident)

    os.makedirs(a.output_dir, exist_ok=True)
    for csv in split_list_across_devices_contiguous(a.example_paths, a.world_size, a.device_id):
        ds = ControlSignalDataset_Balls(base_path=os.path.dirname(csv), metadata_path=csv, is_validation_dataset=True,
                                        num_frames=NUM_FRAMES, height=480, width=832, device=dev)
        ds.min_mass, ds.max_mass, ds.min_force, ds.max_force = 1.0, 4.0, 30.0, 400.0        # INF:137-142
        ds.min_indirect_force, ds.max_indirect_force = ds.min_force, ds.max_force          # INF:145-146
        for i in range(len(ds)):
            data = ds[i]
            extra = {}
            if a.synthetic:
                g = torch.Generator().manual_seed(a.seed)
                for k in ("context_posi", "context_nega"):
                    c = torch.randn((1, 512, 4096), generator=g)
                    c[:, 40:] = 0
                    extra[k] = c.to(torch.bfloat16).to(dev)
            video = pipe(prompt=data["prompt"], negative_prompt=NEGATIVE_PROMPT, input_image=data["video"][0],
                         num_frames=NUM_FRAMES, seed=a.seed, tiled=True, controlnet=True,
                         control_signal_video=data["control_video"], num_inference_steps=a.num_inference_steps, **extra)
            out = os.path.join(a.output_dir, f"{data['file_id']}_seed{a.seed}")
            os.makedirs(out, exist_ok=True)
            for t, frame in enumerate(video):
                frame.save(os.path.join(out, f"{t:03d}.png"))
            print(f"[device {a.device_id}] wrote {len(video)} frames to {out}")


if __name__ == "__main__":
    main()
