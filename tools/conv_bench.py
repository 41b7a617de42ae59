#!/usr/bin/env python3
# NOTE (round 5): the -DCD_WHATIF builds this script can drive are built from the experimental tree (`bash tools/experimental_tree.sh`:
# build/experimental/csrc); the product sources no longer carry those branches.
"""The 96-channel 3x3x3 convolution of one production VAE tile (80 frames of 240 x 416, history in front): direct kernel
(gf_conv_direct.hip) against the implicit GEMM (GF_CONV_DIRECT=0), interleaved in one process; 3.97 TFLOP per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from goal_force_amd import ops

BF = torch.bfloat16


def main():
    T, H, W, C = int(os.environ.get("CONV_T", "80")), 240, 416, 96
    torch.manual_seed(0)
    buf = (torch.randn((T + 2, H, W, C), device="cuda") * 0.7).to(BF)
    x = buf[2:]
    k = 27 * C
    kpad = -(-k // 64) * 64
    w = torch.zeros((C, kpad), dtype=BF, device="cuda")
    w[:, :k] = (torch.randn((C, k), device="cuda") / k ** 0.5).to(BF)
    b = torch.randn((C,), device="cuda").to(BF)
    resid = torch.randn((T * H * W, C), device="cuda").to(BF)
    fl = 2.0 * T * H * W * k * C
    best = {}
    for rnd in range(3):
        for name, env in (("implicit", "0"), ("direct", "1")):
            with ops.options(conv_direct=int(env)):
                for kind, kw in (("bias", {}), ("resid", dict(resid=resid))):
                    ops.vae_conv3d(x, None, w, b, 3, 3, history_in_front=True, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        ops.vae_conv3d(x, None, w, b, 3, 3, history_in_front=True, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    key = f"{name} {kind}"
                    best[key] = min(best.get(key, 1e9), e0.elapsed_time(e1) / 3)
    for key, ms in best.items():
        print(f"{key:16s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    # the 192-channel level of the same tile (half resolution, 3x3x3, 192 -> 192: 3.97 TFLOP): the implicit GEMM
    T2 = T
    b192 = (torch.randn((T2 + 2, H // 2, W // 2, 192), device="cuda") * 0.7).to(BF)
    k192 = 27 * 192
    w192 = (torch.randn((192, k192), device="cuda") / k192 ** 0.5).to(BF)
    bb = torch.randn((192,), device="cuda").to(BF)
    r192 = torch.randn((T2 * (H // 2) * (W // 2), 192), device="cuda").to(BF)
    fl192 = 2.0 * T2 * (H // 2) * (W // 2) * k192 * 192
    for kind, kw in (("bias", {}), ("resid", dict(resid=r192))):
        ops.vae_conv3d(b192[2:], None, w192, bb, 3, 3, history_in_front=True, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.vae_conv3d(b192[2:], None, w192, bb, 3, 3, history_in_front=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"C=192 implicit {kind:6s} {ms:7.3f} ms  {fl192 / ms / 1e9:7.1f} TFLOP/s", flush=True)
    # the decoder's upsample convolution of the same tile (nearest 2x + 3x3, 192 -> 96 channels, source 120 x 208): 2.65 TFLOP
    xs = (torch.randn((T, H // 2, W // 2, 192), device="cuda") * 0.7).to(BF)
    ku = 9 * 192
    wu = (torch.randn((96, ku), device="cuda") / ku ** 0.5).to(BF)
    flu = 2.0 * T * H * W * ku * 96
    bestu = {}
    for rnd in range(3):
        for name, env in (("implicit", "0"), ("direct", "1")):
            with ops.options(conv_direct=int(env)):
                ops.vae_conv3d(xs, None, wu, b, 1, 3, upsample2x=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.vae_conv3d(xs, None, wu, b, 1, 3, upsample2x=True)
                e1.record()
                torch.cuda.synchronize()
                bestu[name] = min(bestu.get(name, 1e9), e0.elapsed_time(e1) / 3)
    for key, ms in bestu.items():
        print(f"upsample {key:8s} {ms:7.3f} ms  {flu / ms / 1e9:7.1f} TFLOP/s", flush=True)
    # timing-only what-if builds of the direct kernel (-DCD_WHATIF=n: 1 no counted wait, 2 no barrier, 4 no weight requests; wrong
    # results): build/whatif/libcd_w<n>.so = gf_conv_direct.hip + gf_gemm.hip + gf_abi.hip
    import ctypes
    import glob
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    out = torch.empty((T * H * W, C), dtype=BF, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    libs = {}
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "whatif", "libcd_w*.so"))):
        lib = ctypes.CDLL(path)
        lib.gf_conv3d_bf16.argtypes = [vp, vp, vp, i64, vp, vp, i64, i64, i64, i64, i64, i64, ci, ci, ci, ci, ci, i64, i64, ci, vp, i64, vp]
        libs[os.path.basename(path)[7:-3]] = lib
    bestw = {}
    for rnd in range(4):                                  # interleaved: the clock drifts over a run
        for name, lib in libs.items():
            def call():
                assert lib.gf_conv3d_bf16(x.data_ptr(), None, w.data_ptr(), kpad, b.data_ptr(), out.data_ptr(), C, T, T, H, W, C, 3, 3, 0, 1, 0,
                                          C, kpad, 0, None, 0, st) == 0
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                call()
            e1.record()
            torch.cuda.synchronize()
            bestw[name] = min(bestw.get(name, 1e9), e0.elapsed_time(e1) / 3)
    for name, ms in bestw.items():
        print(f"what-if {name:12s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s-equivalent", flush=True)


if __name__ == "__main__":
    main()
