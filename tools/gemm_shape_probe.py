#!/usr/bin/env python3
"""Which property of the F->D GEMM costs the per-K-tile time: the long K loop, the 27 KiB row pitch of A / W, or the 20 column tiles?
Times gemm_a4_kernel (and hipBLASLt via F.linear with --ref) on synthetic variants: K and row pitch varied independently."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goal_force_amd import ops  # noqa: E402

S, D, F = 32760, 5120, 13824
BF = torch.bfloat16


def t(fn, n=5, rounds=3):
    best = 1e9
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    ref = "--ref" in sys.argv
    # (label, K, N, pitch of A rows, pitch of W rows)
    cases = [("F->D           K=13824 N=5120  pitch 13824", F, D, F, F),
             ("K short, pitch long K=5120 N=5120  pitch 13824", D, D, F, F),
             ("K long, padded pitch K=13824 N=5120  pitch 13824+64", F, D, F + 64, F + 64),
             ("K long, N wide  K=13824 N=13824 pitch 13824", F, F, F, F),
             ("D->D           K=5120  N=5120  pitch 5120", D, D, D, D),
             ("K=10240 N=5120 pitch 10240", 2 * D, D, 2 * D, 2 * D),
             ("D->F           K=5120  N=13824 pitch 5120", D, F, D, D)]
    for label, k, n, pa, pw in cases:
        xa = torch.randn((S, pa), device="cuda").to(BF)
        wa = (torch.randn((n, pw), device="cuda") * 0.02).to(BF)
        x, w = xa[:, :k], wa[:, :k]
        b = torch.zeros((n,), device="cuda", dtype=BF)
        out = torch.empty((S, n), device="cuda", dtype=BF)
        ms = t(lambda: ops.gemm(x, w, b, out=out))
        line = f"{label:52s} a4 {ms:7.3f} ms {2.0 * S * k * n / ms / 1e9:6.0f} TF  per K tile and round {ms * 1e3 / (k / 64) / (-(-S // 256) * (n // 256) / 256):.3f} us"
        if ref and pa == k:
            msr = t(lambda: torch.nn.functional.linear(x, w, b))
            line += f"   hipBLASLt {msr:7.3f} ms {2.0 * S * k * n / msr / 1e9:6.0f} TF"
        print(line, flush=True)
        del xa, wa, x, w, out


if __name__ == "__main__":
    main()
