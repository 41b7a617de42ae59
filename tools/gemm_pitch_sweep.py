#!/usr/bin/env python3
"""Row-pitch sensitivity of gemm_a4_kernel: K = 5120, N = 5120, M = 32760; the pitch (elements) of A's and W's rows swept."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goal_force_amd import ops  # noqa: E402
from gemm_shape_probe import t  # noqa: E402

S, BF = 32760, torch.bfloat16


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
    n = 5120
    pitches = [int(v) for v in sys.argv[2:]] or [5120, 5184, 5248, 5376, 5632, 6144, 7168, 8192, 9216, 10240, 12288, 13824, 13888, 13952, 14080, 14336, 16384]
    pitches = [p for p in pitches if p >= k]
    b = torch.zeros((n,), device="cuda", dtype=BF)
    out = torch.empty((S, n), device="cuda", dtype=BF)
    for which in ("both", "A only", "W only"):
        for p in pitches:
            pa = p if which != "W only" else k
            pw = p if which != "A only" else k
            xa = torch.randn((S, pa), device="cuda").to(BF)
            wa = (torch.randn((n, pw), device="cuda") * 0.02).to(BF)
            x, w = xa[:, :k], wa[:, :k]
            ms = t(lambda: ops.gemm(x, w, b, out=out), n=6, rounds=3)
            print(f"K={k} pitch {which:6s} {p:6d} el = {2 * p:6d} B: {ms:7.3f} ms {2.0 * S * k * n / ms / 1e9:6.0f} TF", flush=True)
            del xa, wa, x, w


if __name__ == "__main__":
    main()
