#!/usr/bin/env python3
"""What bounds the flash-attention loop: timing-only builds of kernel 2 with parts of the steady phase left out
(-DGF_ATTN_WHATIF=<mask>, see gf_attention.hip; results are wrong by construction, nothing here ships).
  python3 tools/attn_whatif.py --build          (CPU container: cross-compiles build/whatif/libgf_w<mask>.so)
  python3 tools/attn_whatif.py                  (GPU: times every built library on the S = 32760 self-attention)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
OUT = os.path.join(ROOT, "build", "whatif")
MASKS = [0, 1, 2, 4, 8, 16, 3, 7, 15, 31, 63, 5, 62, 64, 67, 128, 256, 143, 271]
NAMES = {1: "no fma/exp/sum/pack", 2: "no max/rescale decision", 4: "no LDS fragment reads", 8: "no wait+barrier",
         16: "no K/V staging", 32: "no deferred O rescale", 128: "no V staging", 256: "no K staging", 64: "(correct results) DMA pieces issued at the top of the phase"}


def describe(m):
    return "shipped loop" if m == 0 else " + ".join(v for k, v in NAMES.items() if m & k)


def build():
    os.makedirs(OUT, exist_ok=True)
    only = [int(a) for a in sys.argv[1:] if a.isdigit()]
    src = [os.path.join(CSRC, f) for f in ("gf_attention.hip", "gf_abi.hip")]
    for m in (only or MASKS):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD",
                        "-mllvm", "-amdgpu-mfma-vgpr-form", f"-DGF_ATTN_WHATIF={m}", f"-I{CSRC}/../include",
                        f"-I{CSRC}", "-o", os.path.join(OUT, f"libgf_w{m}.so")] + src, check=True)
        print("built", m, describe(m), flush=True)


def run():
    import torch
    S, H, D = 32760, 40, 5120
    q = torch.randn((S, D), device="cuda").to(torch.bfloat16)
    k = torch.randn((S, D), device="cuda").to(torch.bfloat16)
    v = torch.randn((S, D), device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    kv_pad = -(-S // 64) * 64
    if os.environ.get("KV_PAD_ODD") and (kv_pad // 64) % 2 == 0:
        kv_pad += 64          # V^T row stride = odd multiple of 128 B: d-rows spread over the memory channels
    vt = torch.zeros((H * 128 * kv_pad,), dtype=torch.bfloat16, device="cuda")
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    st = torch.cuda.current_stream().cuda_stream
    flops = 4.0 * S * S * D
    libs = sorted(glob.glob(os.path.join(OUT, "libgf_w*.so")), key=lambda p: MASKS.index(int(p.split("_w")[-1][:-3])))
    only = [int(a) for a in sys.argv[1:] if a.isdigit()]
    for path in libs:
        m = int(path.split("_w")[-1][:-3])
        if only and m not in only:
            continue
        lib = ctypes.CDLL(path)
        lib.gf_transpose_v.argtypes = [vp, i64, vp, i64, i64, i64, vp]
        lib.gf_flash_attn_fwd_vt.argtypes = [vp] * 5 + [i64] * 8 + [ctypes.c_float, vp]
        assert lib.gf_transpose_v(v.data_ptr(), D, vt.data_ptr(), S, kv_pad, H, st) == 0

        def call():
            rc = lib.gf_flash_attn_fwd_vt(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), None, S, S, kv_pad, H, 128,
                                          D, D, D, 128 ** -0.5, st)
            assert rc == 0
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        print(f"mask {m:2d}  {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s-equivalent   {describe(m)}", flush=True)


if __name__ == "__main__":
    build() if "--build" in sys.argv else run()
