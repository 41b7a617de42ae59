#!/usr/bin/env python3
"""Time the ControlNet training step (goal_force_amd/training.py) at the A14B shape on one MI355X — a measurement tool,
not the driver's bench contract.  One step = training_loss forward (DiT expert + ControlNet) + backward (block-granular
recompute, activation gradients through the frozen expert, weight gradients for the ControlNet) + AdamW.

  python3 tools/train_bench.py --layers 40 --cn-layers 10 --steps 2          # the real model (needs ~120 GB)
  python3 tools/train_bench.py --layers 4 --cn-layers 2 --steps 2            # quick look
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from goal_force_amd import ops, training as tr  # noqa: E402
from goal_force_amd.dit import A14B_CONFIG  # noqa: E402
from goal_force_amd.pipeline import WanVideoPipeline, build_random_controlnet, build_random_expert  # noqa: E402

S, D, F, L = 32760, 5120, 13824, 512


def block_flops():
    return 12.0 * S * D * D + 4.0 * L * D * D + 4.0 * S * D * F + 4.0 * S * S * D + 4.0 * S * L * D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--cn-layers", type=int, default=10)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--frames", type=int, default=21, help="latent frames (21 = 81 video frames)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(A14B_CONFIG)
    cfg["num_layers"] = a.layers
    dit = build_random_expert(cfg, seed=100, device=dev)
    for p in dit.parameters():
        p.requires_grad_(False)
    cn = build_random_controlnet(a.cn_layers, cfg, seed=300, device=dev)
    pipe = WanVideoPipeline.from_modules(dit, None, cn, None, device=dev)
    pipe.scheduler.set_timesteps(1000, training=True)
    g = torch.Generator().manual_seed(0)
    shp = (1, 16, a.frames, 60, 104)
    inp = dict(input_latents=torch.randn(shp, generator=g), noise=torch.randn(shp, generator=g),
               y=torch.randn((1, 20) + shp[2:], generator=g), control=torch.randn(shp, generator=g),
               context=torch.randn((1, L, 4096), generator=g))
    inp = {k: v.to(torch.bfloat16).to(dev) for k, v in inp.items()}
    opt = tr.AdamW(cn.parameters(), lr=1e-5, weight_decay=1e-2)
    times = []
    for step in range(a.steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        with torch.enable_grad():
            loss = tr.training_loss(pipe, input_latents=inp["input_latents"], noise=inp["noise"], context=inp["context"],
                                    y=inp["y"], control_signal_video_latents=inp["control"], timestep_id=500)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step(max_grad_norm=1.0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if step:
            times.append((t1 - t0, t2 - t1, t3 - t2))
        print(f"step {step}: loss {float(loss.detach()):.5f}  fwd {t1 - t0:.3f}s  bwd {t2 - t1:.3f}s  adamw {t3 - t2:.3f}s",
              file=sys.stderr)
    fwd, bwd, upd = (sum(t[i] for t in times) / len(times) for i in range(3))
    nb = a.layers + a.cn_layers
    # forward flops of the blocks; backward = recompute (1x) + activation grads (2x) (+ weight grads 1x GEMM part, ControlNet)
    out = {"metric": "controlnet_training_step_seconds", "value": fwd + bwd + upd, "unit": "s", "fwd_s": fwd, "bwd_s": bwd,
           "adamw_s": upd, "layers": a.layers, "controlnet_layers": a.cn_layers, "tokens": 16 * 0 + a.frames * 30 * 52,
           "forward_block_tflop": nb * block_flops() / 1e12, "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
           "trainable_params": sum(p.numel() for p in cn.parameters())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
