#!/usr/bin/env python3
# NOTE (round 5): the -DGF_LN2_PK=0 build this script can drive are built from the experimental tree (`bash tools/experimental_tree.sh`:
# build/experimental/csrc); the product sources no longer carry those branches.
"""Diagnostic: layernorm_wave2_kernel (pairs, packed fp32 math) against the per-element kernel — where do the two differ, and does
the difference go away when the pair arithmetic is issued as scalar instructions (build/ab/libln_scalar.so, -DGF_LN2_PK=0)?"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goal_force_amd import ops  # noqa: E402

BF = torch.bfloat16
g = torch.Generator().manual_seed(5121)
M, dim = 4000, 5120
x = (torch.randn((M, dim), generator=g) * 3 + 0.5).to(BF).cuda()
a = (1 + 0.3 * torch.randn(dim, generator=g)).to(BF).cuda()
b = (0.4 * torch.randn(dim, generator=g)).to(BF).cuda()
one, zero = torch.ones_like(a), torch.zeros_like(a)


def report(name, new, old):
    d = (new.view(torch.int16).int() - old.view(torch.int16).int()).abs()
    bad = d > 0
    print(f"{name}: mismatches {int(bad.sum())} of {bad.numel()}, max ulp {int(d.max())}, rows affected {int(bad.any(dim=1).sum())}")


olds = {"modulate": ops.layernorm_modulate(x, weight=one, bias=zero, scale1p=a, shift=b),
        "affine": ops.layernorm_modulate(x, weight=a, bias=b, scale1p=one, shift=zero),
        "plain": ops.layernorm_modulate(x, weight=one, bias=zero, scale1p=one, shift=zero)}
report("packed modulate", ops.layernorm_modulate(x, scale1p=a, shift=b), olds["modulate"])
report("packed affine  ", ops.layernorm_modulate(x, weight=a, bias=b), olds["affine"])
report("packed plain   ", ops.layernorm_modulate(x), olds["plain"])
path = os.path.join(ROOT, "build", "ab", "libln_scalar.so")
if os.path.exists(path):
    lib = ctypes.CDLL(path)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.gf_layernorm_modulate.argtypes = [vp] * 6 + [i64] * 4 + [ctypes.c_float, vp]
    st = torch.cuda.current_stream().cuda_stream

    def scalar(w=None, bi=None, sc=None, sh=None):
        out = torch.empty_like(x)
        p = lambda t: None if t is None else t.data_ptr()
        assert lib.gf_layernorm_modulate(x.data_ptr(), out.data_ptr(), p(w), p(bi), p(sc), p(sh), M, dim, dim, dim, 1e-6, st) == 0
        return out
    report("scalar modulate", scalar(sc=a, sh=b), olds["modulate"])
    report("scalar affine  ", scalar(w=a, bi=b), olds["affine"])
    report("scalar plain   ", scalar(), olds["plain"])
ref = torch.nn.functional.layer_norm(x.float(), (dim,), eps=1e-6).to(BF)
print("vs torch fp32 layer_norm -> bf16: packed", int((ops.layernorm_modulate(x) != ref).sum()), "per-element", int((olds["plain"] != ref).sum()))
