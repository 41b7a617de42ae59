#!/usr/bin/env python3
"""K-loop start stagger of gemm_a4_kernel: time the three block shapes for several GF_A4_STAGGER values in one process (interleaved)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goal_force_amd import ops  # noqa: E402

S, D, F = 32760, 5120, 13824
BF = torch.bfloat16


def main():
    vals = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 5, 7, 11]
    shapes = {"D->D": (D, D), "D->F": (D, F), "F->D": (F, D)}
    for name, (k, n) in shapes.items():
        x = torch.randn((S, k), device="cuda").to(BF)
        w = (torch.randn((n, k), device="cuda") * 0.02).to(BF)
        b = torch.zeros((n,), device="cuda", dtype=BF)
        out = torch.empty((S, n), device="cuda", dtype=BF)
        best = {v: 1e9 for v in vals}
        for rnd in range(4):
            for v in vals:
                ops.options(a4_stagger=v).__enter__()      # (left set: the next value overwrites it)
                ops.gemm(x, w, b, out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm(x, w, b, out=out)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / 5)
        print(name, "  ".join(f"s={v}: {ms:.3f} ms {2.0 * S * k * n / ms / 1e9:.0f} TF" for v, ms in best.items()), flush=True)


if __name__ == "__main__":
    main()
