#!/usr/bin/env python3
"""FFN1 (GELU epilogue) followed by FFN2 (gate*+resid epilogue) as a pair, per library build in build/ab/libgemm_*.so: does the way
FFN1 writes its 906 MB output (non-temporal or not) change what FFN2 pays to read it?"""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    libs = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "build", "ab", "libgemm_*.so"))):
        lib = ctypes.CDLL(path)
        lib.gf_gemm_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp, i64, vp, vp]
        libs[os.path.basename(path)[8:-3]] = lib
    S, D, F, BF = 32760, 5120, 13824, torch.bfloat16
    st = torch.cuda.current_stream().cuda_stream
    h = torch.randn((S, D), device="cuda").to(BF)
    w1 = (torch.randn((F, D), device="cuda") * 0.02).to(BF)
    w2 = (torch.randn((D, F), device="cuda") * 0.02).to(BF)
    b1, b2 = torch.zeros((F,), device="cuda", dtype=BF), torch.zeros((D,), device="cuda", dtype=BF)
    res = torch.randn((S, D), device="cuda").to(BF)
    gate = torch.randn((D,), device="cuda").to(BF)
    f1 = torch.empty((S, F), device="cuda", dtype=BF)
    out = torch.empty((S, D), device="cuda", dtype=BF)
    best = {n: [1e9, 1e9, 1e9] for n in libs}
    for rnd in range(5):
        for name, lib in libs.items():
            def ffn1():
                assert lib.gf_gemm_bf16(h.data_ptr(), D, w1.data_ptr(), D, b1.data_ptr(), f1.data_ptr(), F, S, F, D, 1, None, 0, None, st) == 0

            def ffn2():
                assert lib.gf_gemm_bf16(f1.data_ptr(), F, w2.data_ptr(), F, b2.data_ptr(), out.data_ptr(), D, S, D, F, 2, res.data_ptr(), D, gate.data_ptr(), st) == 0
            ffn1(); ffn2()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            t1 = t2 = 0.0
            for _ in range(4):
                ev[0].record(); ffn1(); ev[1].record(); ffn2(); ev[2].record()
                torch.cuda.synchronize()
                t1 += ev[0].elapsed_time(ev[1]) / 4
                t2 += ev[1].elapsed_time(ev[2]) / 4
            b = best[name]
            b[0], b[1], b[2] = min(b[0], t1), min(b[1], t2), min(b[2], t1 + t2)
    for name, (t1, t2, tt) in best.items():
        print(f"{name:8s} FFN1 {t1:.3f} ms  FFN2 {t2:.3f} ms  pair {tt:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
