#!/usr/bin/env python3
"""Diagnostic: builds the phased GEMM with in-kernel s_memtime stamps (GF_GEMM_STAMP=1) into a separate library under
gpurun_out/ and prints where waves 0 (row 0) and 4 (row 1) of workgroup 0 spend their cycles per K-tile phase.
Shares are what matter; the stamped build is slower than the shipped one (the stamps fence the scheduler)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
sys.path.insert(0, ROOT)
import torch

out = os.path.join(ROOT, "gpurun_out", "libgf_gemm_stamp.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
src = [os.path.join(CSRC, f) for f in ("gf_gemm.hip", "gf_abi.hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD",
                "-DGF_GEMM_STAMP=1", f"-I{CSRC}/../include", f"-I{CSRC}", "-o", out] + src, check=True)
lib = ctypes.CDLL(out)
M, N, K = 32760, 5120, 5120
a = torch.randn((M, K), device="cuda").to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16)
c = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.gf_debug_set_gemm_buffer(ctypes.c_void_p(dbg.data_ptr()))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.gf_gemm_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp, i64, vp, vp]
for _ in range(3):
    rc = lib.gf_gemm_bf16(a.data_ptr(), K, w.data_ptr(), K, None, c.data_ptr(), N, M, N, K, 0, None, 0, None,
                          torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
d = dbg.cpu().tolist()
nph = 4 * (K // 64)
names = ["DMA issue (2 pieces)", "ds_read issue", "s_waitcnt vmcnt(8)", "s_waitcnt lgkmcnt(0)", "barrier (partner still computing)",
         "MFMA burst (16 x 16x16x32)", "barrier (partner still loading)", "-"]
for wv, label in ((0, "wave 0 (row 0)"), (1, "wave 4 (row 1)")):
    tot = sum(d[wv * 8:(wv + 1) * 8])
    print(f"{label}: {tot / nph:.0f} cycles per phase ({4 * tot / nph:.0f} per K-tile; 2048 = MFMA-bound)")
    for i in range(7):
        print(f"   seg {i} {names[i]:36s} {d[wv * 8 + i] / nph:8.0f} cyc  {100 * d[wv * 8 + i] / max(tot, 1):5.1f} %")
