#!/usr/bin/env python3
"""The 3x3x3 convolutions of the 192- / 384-channel VAE levels at the shapes of a production tile (81 frames of a 240 x 416 tile):
the padded-layout kernel (gf_conv_a4.hip) against the implicit GEMM (gf_conv3d_bf16), interleaved in one process.

    python tools/conv_a4_bench.py [frames at the 120x208 level, default 81]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from goal_force_amd import ops  # noqa: E402

BF = torch.bfloat16


def one(T, H, W, C, N, rounds=3, reps=3):
    k = 27 * C
    src = (torch.randn((T + 2, H, W, C), device="cuda") * 0.7).to(BF)
    w = (torch.randn((N, k), device="cuda") / k ** 0.5).to(BF)
    b = torch.randn((N,), device="cuda").to(BF)
    resid = torch.randn((T * H * W, N), device="cuda").to(BF)
    buf, hist, cur = ops.padded_activation(T, H, W, C, "cuda")
    hist.copy_(src[:2])
    cur.copy_(src[2:])
    fl = 2.0 * T * H * W * k * N
    best = {}
    for rnd in range(rounds):
        for name in ("implicit", "padded"):
            for kind, kw in (("bias", {}), ("resid", dict(resid=resid))):
                if name == "implicit":
                    with ops.options(conv_direct=0):
                        fn = lambda: ops.vae_conv3d(src[2:], None, w, b, 3, 3, history_in_front=True, **kw)   # noqa: E731
                        fn()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(reps):
                            fn()
                        e1.record()
                else:
                    fn = lambda: ops.vae_conv3d_padded(buf, w, b, **kw)   # noqa: E731
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                torch.cuda.synchronize()
                key = (name, kind)
                best[key] = min(best.get(key, 1e9), e0.elapsed_time(e1) / reps)
    same = torch.equal(ops.vae_conv3d_padded(buf, w, b), ops.vae_conv3d(src[2:], None, w, b, 3, 3, history_in_front=True))
    for (name, kind), ms in sorted(best.items()):
        print(f"{C:3d}->{N:3d} {T:2d}x{H:3d}x{W:3d} {name:8s} {kind:5s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    print(f"   bit-identical: {same}", flush=True)


def up(T, Hs, Ws, C=384, N=192, rounds=3, reps=3):
    """the resample convolution behind the 2x upsample: folded gather (implicit GEMM) against upsample-into-padded + kt = 1 kernel"""
    k = 9 * C
    x = (torch.randn((T, Hs, Ws, C), device="cuda") * 0.7).to(BF)
    w = (torch.randn((N, k), device="cuda") / k ** 0.5).to(BF)
    b = torch.randn((N,), device="cuda").to(BF)
    buf = ops.padded_activation(T, 2 * Hs, 2 * Ws, C, "cuda", history=False)[0]
    fl = 2.0 * T * 4 * Hs * Ws * k * N
    fns = {"implicit": lambda: ops.vae_conv3d(x, None, w, b, 1, 3, upsample2x=True),
           "padded": lambda: (ops.vae_upsample2x_padded(x, buf), ops.vae_conv3d_padded(buf, w, b, kt=1))[1]}
    best = {}
    for rnd in range(rounds):
        for name, fn in fns.items():
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / reps)
    same = torch.equal(fns["padded"](), fns["implicit"]())
    for name, ms in sorted(best.items()):
        print(f"up2x+3x3 {C}->{N} {T:2d}x{2 * Hs:3d}x{2 * Ws:3d} {name:8s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s (upsample pass included)", flush=True)
    print(f"   bit-identical: {same}", flush=True)


def main():
    T2 = int(sys.argv[1]) if len(sys.argv) > 1 else 81
    one(T2, 120, 208, 192, 192)                       # decoder level 2: 6 of these per tile (4.02 TFLOP each at 81 frames)
    one((T2 + 1) // 2, 60, 104, 384, 384)             # decoder level 1: 5 per tile
    one((T2 + 1) // 2, 60, 104, 192, 384)             # ... and its first convolution
    one((T2 + 3) // 4, 30, 52, 384, 384)              # level 0 / middle: 10 per tile
    up(T2, 60, 104)                                   # the resample convolution of level 1 -> 2 (2.68 TFLOP)
    up((T2 + 1) // 2, 30, 52)                         # ... of level 0 -> 1


if __name__ == "__main__":
    main()
