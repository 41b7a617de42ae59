#!/usr/bin/env python3
# NOTE (round 5): this script drives variants that are no longer in the product library (GF_* environment selectors, v1 / sl
# kernels, what-if builds).  It runs against a library built from the experimental tree: `bash tools/experimental_tree.sh`, then build
# build/experimental/csrc as the Makefile builds goal_force_amd/csrc and point GOALFORCE_HIP_LIB at the result.
"""A/B of the two bf16 K loops of gemm_a4_kernel in ONE process (interleaved rounds): the k-sub-step loop (three barriers per K
tile, tools/gen_gemm_a4.py) against the half-tile loop (two barriers, staging by halves: the fp8 loop's schedule with bf16 MFMAs,
A4F8_BF16=1 tools/gen_gemm_a4f8.py), selected per launch with GF_A4_LOOP.  Checks that both produce the same bits.
Needs a library built with the second loop compiled in (the shipped one carries only the first):
    A4F8_BF16=1 A4F8_OUT=build/ab/a4h_loop.inc python tools/gen_gemm_a4f8.py
    make -C goal_force_amd/csrc CXXFLAGS+='-DGF_A4_HALFTILE_AB=\"$PWD/build/ab/a4h_loop.inc\"'     (then rebuild plainly)
Result (profiles/r03/gemm_loop_ab_halftile.log): bit-identical, and within 0.5 % on all three shapes — not shipped."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from goal_force_amd import ops

S, D, F = 32760, 5120, 13824
BF = torch.bfloat16


def main():
    rounds = int(os.environ.get("GEMM_AB_ROUNDS", "6"))
    torch.manual_seed(0)
    for name, (n, k) in {"D->D": (D, D), "D->F": (F, D), "F->D": (D, F)}.items():
        x = torch.randn((S, k), device="cuda").to(BF)
        w = (torch.randn((n, k), device="cuda") / k ** 0.5).to(BF)
        b = torch.randn((n,), device="cuda").to(BF)
        out = torch.empty((S, n), device="cuda", dtype=BF)
        fl = 2.0 * S * n * k
        best, outs = {}, {}
        variants = [("ksub", None), ("half", "h"), ("torch", None)]
        for rnd in range(rounds):
            for vn, env in variants:
                os.environ.pop("GF_A4_LOOP", None)
                ops.reload_options()       # the launch paths read the knobs once per process
                if env:
                    os.environ["GF_A4_LOOP"] = env
                    ops.reload_options()       # the launch paths read the knobs once per process
                call = (lambda: torch.nn.functional.linear(x, w, b)) if vn == "torch" else (lambda: ops.gemm(x, w, b, out=out))
                for _ in range(2):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best[vn] = min(best.get(vn, 1e9), e0.elapsed_time(e1) / 6)
                if vn != "torch" and rnd == 0:
                    outs[vn] = out.clone()
        os.environ.pop("GF_A4_LOOP", None)
        ops.reload_options()       # the launch paths read the knobs once per process
        same = torch.equal(outs["ksub"], outs["half"])
        for vn, ms in best.items():
            print(f"{name}  {vn:6s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s" + ("" if vn == "torch" else f"  bits {'==' if same else 'DIFFER'}"), flush=True)


if __name__ == "__main__":
    main()
