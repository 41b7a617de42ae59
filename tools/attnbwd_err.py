#!/usr/bin/env python3
# NOTE (round 5): this script drives variants that are no longer in the product library (GF_* environment selectors, v1 / sl
# kernels, what-if builds).  It runs against a library built from the experimental tree: `bash tools/experimental_tree.sh`, then build
# build/experimental/csrc as the Makefile builds goal_force_amd/csrc and point GOALFORCE_HIP_LIB at the result.
"""Flash-attention backward against an fp64 autograd reference and against the first kernels (GF_ATTN_BWD=v1) at two mid sizes:
rel-L2 of dq / dk / dv (what a change of the P arithmetic costs in accuracy; run on the GPU)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goal_force_amd import ops
BF = torch.bfloat16; HD = 128
def rel(a, b): return float((a - b).norm() / b.norm())
for sq, skv, heads in [(1000, 777, 8), (2048, 2048, 8)]:
    g = torch.Generator().manual_seed(sq + skv)
    q, k, v, do = (torch.randn((n, heads * HD), generator=g).to(BF).cuda() for n in (sq, skv, skv, sq))
    qf, kf, vf = (t.double().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(-1, heads, HD).transpose(0, 1) for t in (qf, kf, vf))
    s = qh @ kh.transpose(1, 2) / math.sqrt(HD)
    o_ref = (torch.softmax(s, -1) @ vh).transpose(0, 1).reshape(sq, heads * HD)
    o_ref.backward(do.double())
    o, lse = ops.flash_attn_lse(q, k, v, heads)
    new = ops.flash_attn_bwd(q, k, v, o, do, lse, heads)
    with ops.env_options(GF_ATTN_BWD="v1"):
        old = ops.flash_attn_bwd(q, k, v, o, do, lse, heads)
    for name, a, b, r in zip(("dq", "dk", "dv"), new, old, (qf.grad, kf.grad, vf.grad)):
        print(f"S={sq}x{skv} {name}: new vs fp64 {rel(a.double(), r):.3e}   v1 vs fp64 {rel(b.double(), r):.3e}   new vs v1 {rel(a.double(), b.double()):.3e}", flush=True)
