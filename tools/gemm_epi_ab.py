#!/usr/bin/env python3
"""Two library builds (build/ab/libgemm_*.so) on the block's GEMMs WITH their epilogues: time and bits (gf_gemm_bf16 through ctypes)."""
import ctypes
import glob
import os

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
    res = torch.randn((S, D), device="cuda").to(BF)
    gate = torch.randn((D,), device="cuda").to(BF)
    cases = {"D->D bias": (D, D, 0), "D->D gate*+resid": (D, D, 2), "D->D +resid": (D, D, 3), "D->F bias": (D, F, 0), "D->F GELU": (D, F, 1), "F->D bias": (F, D, 0), "F->D gate*+resid": (F, D, 2)}
    for name, (k, n, epi) in cases.items():
        x = torch.randn((S, k), device="cuda").to(BF)
        w = (torch.randn((n, k), device="cuda") * 0.02).to(BF)
        b = (torch.randn((n,), device="cuda") * 0.1).to(BF)
        best, outs = {}, {}
        for rnd in range(6):
            for ln, lib in libs.items():
                out = torch.empty((S, n), device="cuda", dtype=BF)
                call = lambda: lib.gf_gemm_bf16(x.data_ptr(), k, w.data_ptr(), k, b.data_ptr(), out.data_ptr(), n, S, n, k, epi,
                                                res.data_ptr() if epi in (2, 3) else None, D, gate.data_ptr() if epi == 2 else None, st)
                assert call() == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best[ln] = min(best.get(ln, 1e9), e0.elapsed_time(e1) / 5)
                outs[ln] = out
        first = next(iter(outs.values()))
        print(f"{name:18s}", "  ".join(f"{ln}: {ms:.3f} ms {2.0 * S * k * n / ms / 1e9:.0f} TF{'' if torch.equal(outs[ln], first) else ' DIFFERS'}" for ln, ms in best.items()), flush=True)


if __name__ == "__main__":
    main()
