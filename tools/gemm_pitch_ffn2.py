#!/usr/bin/env python3
"""F->D (FFN2) with the A operand's row pitch padded: interleaved rounds in one process.  A = [32760, 13824] inside rows of `pitch` elements."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goal_force_amd import ops  # noqa: E402

S, D, F, BF = 32760, 5120, 13824, torch.bfloat16


def main():
    pitches = [int(v) for v in sys.argv[1:]] or [13824, 13888, 13952, 14080, 14336, 14848, 15360, 16384]
    w = (torch.randn((D, F), device="cuda") * 0.02).to(BF)
    b = torch.zeros((D,), device="cuda", dtype=BF)
    out = torch.empty((S, D), device="cuda", dtype=BF)
    res = torch.randn((S, D), device="cuda").to(BF)
    gate = torch.randn((D,), device="cuda").to(BF)
    xs = {p: torch.randn((S, p), device="cuda").to(BF)[:, :F] for p in pitches}
    best = {p: 1e9 for p in pitches}
    beste = {p: 1e9 for p in pitches}
    for rnd in range(5):
        for p in pitches:
            for epi in (0, 1):
                fn = (lambda: ops.gemm(xs[p], w, b, out=out)) if epi == 0 else \
                     (lambda: ops.gemm(xs[p], w, b, out=out, epilogue=ops.EPI_BIAS_GATE_RESID, resid=res, gate=gate))
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 4
                if epi:
                    beste[p] = min(beste[p], ms)
                else:
                    best[p] = min(best[p], ms)
    fl = 2.0 * S * F * D
    for p in pitches:
        print(f"A pitch {p:6d} el = {2 * p:6d} B: plain {best[p]:6.3f} ms {fl / best[p] / 1e9:6.0f} TF   gate*+resid {beste[p]:6.3f} ms {fl / beste[p] / 1e9:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
