#!/usr/bin/env python3
# NOTE (round 5): this script drives variants that are no longer in the product library (GF_* environment selectors, v1 / sl
# kernels, what-if builds).  It runs against a library built from the experimental tree: `bash tools/experimental_tree.sh`, then build
# build/experimental/csrc as the Makefile builds goal_force_amd/csrc and point GOALFORCE_HIP_LIB at the result.
"""A/B of GEMM library builds in ONE process (interleaved rounds, same device): every build/ab/libgemm_*.so, each with
GF_GEMM_KERNEL = a4 and ph, on the three DiT shapes (+ torch F.linear = hipBLASLt as the yardstick)."""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    libs = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "build", "ab", "libgemm_*.so"))):
        lib = ctypes.CDLL(path)
        lib.gf_gemm_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp, i64, vp, vp]
        libs[os.path.basename(path)[8:-3]] = lib
    S, D, F = 32760, 5120, 13824
    st = torch.cuda.current_stream().cuda_stream
    for name, (n, k) in {"D->D": (D, D), "D->F": (F, D), "F->D": (D, F)}.items():
        x = torch.randn((S, k), device="cuda").to(torch.bfloat16)
        w = (torch.randn((n, k), device="cuda") / k ** 0.5).to(torch.bfloat16)
        out = torch.empty((S, n), device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * S * n * k
        kerns = os.environ.get("GEMM_AB_KERNELS", "a4,ph").split(",")
        variants = [(ln, kern) for ln in libs for kern in kerns] + [("torch", "F.linear")]
        best, outs = {}, {}
        for rnd in range(int(os.environ.get("GEMM_AB_ROUNDS", "4"))):
            for ln, kern in variants:
                if ln == "torch":
                    call = lambda: torch.nn.functional.linear(x, w)
                else:
                    os.environ["GF_GEMM_KERNEL"] = kern
                    libs[ln].gf_reload_options()       # the library reads its knobs once; tell it the environment changed
                    lib = libs[ln]
                    call = lambda: lib.gf_gemm_bf16(x.data_ptr(), k, w.data_ptr(), k, None, out.data_ptr(), n, S, n, k, 0, None, 0, None, st)
                for _ in range(2):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best[(ln, kern)] = min(best.get((ln, kern), 1e9), e0.elapsed_time(e1) / 6)
                if ln != "torch" and rnd == 0:
                    outs[(ln, kern)] = out.clone()
        first = next(iter(outs.values()))
        for (ln, kern), ms in best.items():
            same = "" if ln == "torch" else ("  == first" if torch.equal(outs[(ln, kern)], first) else "  DIFFERS from first")
            print(f"{name}  {ln:10s} {kern:9s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s{same}", flush=True)


if __name__ == "__main__":
    main()
