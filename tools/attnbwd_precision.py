#!/usr/bin/env python3
"""Attention forward (lse variant) + backward against fp64 autograd at unit and PEAKY logits, beside torch-ROCm's bf16 SDPA under autograd
(what the reference's training step runs): rel-L2 of o, dq, dk, dv.

  plain q       the kernels on a bf16 q with scale 1/sqrt(d): the forward rounds Q' = bf16(q c) a second time, the backward rebuilds P
                from the once-rounded q and the forward's log-sum-exp (rounds 1-5's training path)
  pre-scaled q  q' = bf16(c q) made from the fp32 q with ONE rounding (the module: RopeTable.scaled), scale = ln 2 in forward and
                backward: the same scores on both sides (round 6); its fp64 reference is evaluated on q' / c, its dq is c dq'

`python tools/attnbwd_precision.py [--s 2048] [--skv 2048] [--heads 8]`."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from goal_force_amd import ops  # noqa: E402
from goal_force_amd.dit import Q_PRESCALE  # noqa: E402

BF, HD = torch.bfloat16, 128


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def fp64_grads(q, k, v, do, H):
    S, SK = q.shape[0], k.shape[0]
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    with torch.enable_grad():
        p = torch.softmax(q64.view(S, H, HD).transpose(0, 1) @ k64.view(SK, H, HD).permute(1, 2, 0) / math.sqrt(HD), -1)
        o64 = (p @ v64.view(SK, H, HD).transpose(0, 1)).transpose(0, 1).reshape(S, H * HD)
        o64.backward(do.double())
    return o64.detach(), q64.grad, k64.grad, v64.grad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--s", type=int, default=2048)
    ap.add_argument("--skv", type=int, default=None)
    ap.add_argument("--heads", type=int, default=8)
    a = ap.parse_args()
    S, H = a.s, a.heads
    SK = a.skv or S
    c = Q_PRESCALE(HD)
    g = torch.Generator(device="cuda").manual_seed(0)
    row = lambda t, ref: "  ".join(f"{n} {rel(x, r):.2e}" for n, x, r in zip(("o", "dq", "dk", "dv"), t, ref))
    for qs in (1.0, 3.0, 8.0):
        q32 = torch.randn((S, H * HD), generator=g, device="cuda") * qs
        k, v = (torch.randn((SK, H * HD), generator=g, device="cuda").to(BF) for _ in range(2))
        do = torch.randn((S, H * HD), generator=g, device="cuda").to(BF)
        q = q32.to(BF)
        ref = fp64_grads(q, k, v, do, H)
        # torch-ROCm SDPA, bf16, autograd
        qb, kb, vb = (t.clone().requires_grad_(True) for t in (q, k, v))
        with torch.enable_grad():
            ob = F.scaled_dot_product_attention(qb.view(S, H, HD).transpose(0, 1)[None], kb.view(SK, H, HD).transpose(0, 1)[None],
                                                vb.view(SK, H, HD).transpose(0, 1)[None])[0].transpose(0, 1).reshape(S, H * HD)
            ob.backward(do)
        sd = (ob.detach(), qb.grad, kb.grad, vb.grad)
        # HIP, plain q
        o, lse = ops.flash_attn_lse(q, k, v, H)
        plain = (o,) + tuple(ops.flash_attn_bwd(q, k, v, o, do, lse, H))
        # HIP, pre-scaled q
        qp = (q32 * c).to(BF)
        refp = fp64_grads(qp.double() / c, k, v, do, H)
        o, lse = ops.flash_attn_lse(qp, k, v, H, scale=math.log(2.0))
        dqp, dk, dv = ops.flash_attn_bwd(qp, k, v, o, do, lse, H, scale=math.log(2.0))
        pre = (o, dqp.double() * c, dk, dv)
        print(f"Sq={S} Skv={SK} heads={H} logit std {qs:g}:\n    HIP plain q       {row(plain, ref)}\n    HIP pre-scaled q  {row(pre, refp)}\n"
              f"    torch SDPA bf16   {row(sd, ref)}", flush=True)


if __name__ == "__main__":
    main()
