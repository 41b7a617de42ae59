#!/usr/bin/env python3
"""Row-wise kernels: a previous library build (build/ab/librow_prev.so) against the current one — same bits? how fast?"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    lib = ctypes.CDLL(path)
    vp, i64, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
    lib.gf_layernorm_modulate.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, vp]
    lib.gf_rmsnorm_rope.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, f32, vp]
    return lib


def main():
    libs = {"prev": load(os.path.join(ROOT, "build", "ab", "librow_prev.so")), "new": load(os.path.join(ROOT, "goal_force_amd", "libgoalforce_hip.so"))}
    S, D, BF = 32760, 5120, torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    x = (torch.randn((S, D), device="cuda") * 2 + 0.3).to(BF)
    w, b, sc, sh = ((torch.randn((D,), device="cuda") * 0.3 + o).to(BF) for o in (1, 0, 1, 0))
    cos = torch.rand((S, 64), device="cuda")
    sin = torch.rand((S, 64), device="cuda")
    p = lambda t: None if t is None else t.data_ptr()
    cases = {"LN plain": (None, None, None, None), "LN affine": (w, b, None, None), "LN modulate": (None, None, sc, sh), "LN weight only": (w, None, None, None)}
    for name, (cw, cb, csc, csh) in cases.items():
        outs, times = {}, {}
        for ln, lib in libs.items():
            out = torch.empty_like(x)
            call = lambda: lib.gf_layernorm_modulate(p(x), p(out), p(cw), p(cb), p(csc), p(csh), S, D, D, D, 1e-6, st)
            assert call() == 0
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            outs[ln], times[ln] = out, best
        print(f"{name:16s} prev {times['prev']:.4f} ms  new {times['new']:.4f} ms  ({2 * x.numel() * 2 / times['new'] / 1e9:.2f} TB/s)  same bits: {torch.equal(outs['prev'], outs['new'])}")
    for name, (c, s_) in {"RMSNorm+RoPE": (cos, sin), "RMSNorm": (None, None)}.items():
        outs, times = {}, {}
        for ln, lib in libs.items():
            y = x.clone()
            assert lib.gf_rmsnorm_rope(p(y), p(w), p(c), p(s_), S, D, 128, D, 1e-6, st) == 0
            outs[ln] = y.clone()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    lib.gf_rmsnorm_rope(p(y), p(w), p(c), p(s_), S, D, 128, D, 1e-6, st)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            times[ln] = best
        print(f"{name:16s} prev {times['prev']:.4f} ms  new {times['new']:.4f} ms  ({2 * x.numel() * 2 / times['new'] / 1e9:.2f} TB/s)  same bits: {torch.equal(outs['prev'], outs['new'])}")


if __name__ == "__main__":
    main()
