#!/usr/bin/env python3
"""GF_A4_GROUP_M (row tiles per workgroup-order group of gemm_a4_kernel) on the block's GEMM shapes with their in-loop epilogues, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goal_force_amd import ops  # noqa: E402

S, D, F, BF = 32760, 5120, 13824, torch.bfloat16


def main():
    vals = [int(v) for v in sys.argv[1:]] or [8, 4, 2]
    res = torch.randn((S, D), device="cuda").to(BF)
    gate = torch.randn((D,), device="cuda").to(BF)
    cases = {"D->D bias": (D, D, ops.EPI_BIAS), "D->D gate*+resid": (D, D, ops.EPI_BIAS_GATE_RESID), "D->F GELU": (D, F, ops.EPI_BIAS_GELU_TANH),
             "F->D gate*+resid": (F, D, ops.EPI_BIAS_GATE_RESID)}
    for name, (k, n, epi) in cases.items():
        x = torch.randn((S, k), device="cuda").to(BF)
        w = (torch.randn((n, k), device="cuda") * 0.02).to(BF)
        b = torch.zeros((n,), device="cuda", dtype=BF)
        out = torch.empty((S, n), device="cuda", dtype=BF)
        kw = dict(epilogue=epi)
        if epi == ops.EPI_BIAS_GATE_RESID:
            kw.update(resid=res, gate=gate)
        best = {v: 1e9 for v in vals}
        for rnd in range(5):
            for v in vals:
                ops.options(a4_group_m=v).__enter__()      # (left set: the next value overwrites it; reset after the sweep)
                ops.gemm(x, w, b, out=out, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm(x, w, b, out=out, **kw)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / 5)
        print(f"{name:18s}", "  ".join(f"group_m={v}: {ms:.3f} ms {2.0 * S * k * n / ms / 1e9:.0f} TF" for v, ms in best.items()), flush=True)


if __name__ == "__main__":
    main()
