#!/usr/bin/env python3
"""Diagnostic: builds the flash-attention kernel with in-kernel s_memtime stamps (GF_ATTN_STAMP=1) into a separate
library under gpurun_out/ and prints where waves 0 and 4 of workgroup 0 spend their cycles per KV tile.
Shares are what matter; the stamped build is slower than the shipped one (fences forbid overlaps)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
sys.path.insert(0, ROOT)
import torch

out = os.path.join(ROOT, "gpurun_out", "libgf_stamp.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
src = [os.path.join(CSRC, f) for f in ("gf_attention.hip", "gf_abi.hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD",
                "-DGF_ATTN_STAMP=1", f"-I{CSRC}/../include", f"-I{CSRC}", "-o", out] + src, check=True)
lib = ctypes.CDLL(out)
S, H, D = 32760, 40, 5120
q = torch.randn((S, D), device="cuda").to(torch.bfloat16)
k = torch.randn((S, D), device="cuda").to(torch.bfloat16)
v = torch.randn((S, D), device="cuda").to(torch.bfloat16)
o = torch.empty_like(q)
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.gf_debug_set_attn_buffer(ctypes.c_void_p(dbg.data_ptr()))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.gf_flash_attn_fwd.argtypes = [vp] * 4 + [i64] * 8 + [ctypes.c_float, vp]
for _ in range(3):
    rc = lib.gf_flash_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), S, S, H, 128, D, D, D, D,
                               128 ** -0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
d = dbg.cpu().tolist()
nt = (S + 63) // 64
names = ["barrier->top", "QK^T (+mask)", "max/shuffle/decide", "exp/sum/cvt", "PV (early: after softmax; late: before QK)",
         "wait loads + ds_write", "barrier", "-"]
for w, label in ((0, "wave 0 (early)"), (1, "wave 4 (late)")):
    tot = sum(d[w * 8:(w + 1) * 8])
    print(f"{label}: {tot / nt:.0f} cycles per KV tile")
    for i in range(7):
        print(f"   seg {i} {names[i]:45s} {d[w * 8 + i] / nt:8.0f} cyc  {100 * d[w * 8 + i] / max(tot, 1):5.1f} %")
