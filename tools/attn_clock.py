#!/usr/bin/env python3
"""In-kernel clock of the self-attention kernel (flash_attn_fwd_kernel3) under its own load — MI355X_MICROARCH.md, DVFS item 6.

  python3 tools/attn_clock.py --build        (CPU container: build/ab/gfclock.so = gf_attention.hip with -DGF_K3_CLOCK=1)
  python3 tools/attn_clock.py [--zeros]      (GPU: >= 2 s of back-to-back launches on random data, then one stamped launch)

The stamped build reads s_memtime (shader cycles) and s_memrealtime (100 MHz) once before and once after the steady loop of every
workgroup; clock = d(memtime) / d(memrealtime) x 100 MHz, median over the 5120 workgroups.  Printed beside it: launch time by HIP
events, and the MFMA-busy fraction the launch would have at that clock if its 21.98 TFLOP + row-sum MFMAs were all the matrix
pipe did (16 cycles per v_mfma_f32_16x16x32_bf16 per SIMD)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
OUT = os.path.join(ROOT, "build", "ab", "gfclock.so")


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = [os.path.join(CSRC, f) for f in ("gf_attention.hip", "gf_abi.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD", "-DGF_K3_CLOCK=1",
                    "-mllvm", "-amdgpu-mfma-vgpr-form", f"-I{CSRC}/../include", f"-I{CSRC}", "-o", OUT] + src, check=True)
    print("built", OUT)


def run():
    import torch
    s, H, D = 32760, 40, 5120
    lib = ctypes.CDLL(OUT)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.gf_transpose_v32.argtypes = [vp, i64, vp, i64, i64, i64, vp]
    lib.gf_flash_attn_fwd_vt32.argtypes = [vp] * 5 + [i64] * 8 + [ctypes.c_float, vp]
    lib.gf_debug_set_attn_buffer.argtypes = [vp]
    q, k, v = (torch.randn((s, D), device="cuda").to(torch.bfloat16) for _ in range(3))
    if "--zeros" in sys.argv:
        q, k, v = (torch.zeros_like(t) for t in (q, k, v))
    o = torch.empty_like(q)
    kv_pad = -(-s // 64) * 64
    vt = torch.zeros((H * 128 * kv_pad,), dtype=torch.bfloat16, device="cuda")
    nwg = -(-s // 256) * H
    dbg = torch.zeros((2 * nwg,), dtype=torch.int64, device="cuda")
    lib.gf_debug_set_attn_buffer(dbg.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    assert lib.gf_transpose_v32(v.data_ptr(), D, vt.data_ptr(), s, kv_pad, H, st) == 0

    def call():
        assert lib.gf_flash_attn_fwd_vt32(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), None, s, s, kv_pad, H, 128, D, D, D, 128 ** -0.5, st) == 0

    n = 160                                   # ~2.5 s of back-to-back launches: the clock has settled when the last one runs
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(n):
        if i == n - 8:
            e0.record()
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    d = dbg.cpu().view(nwg, 2).double()
    ok = d[:, 1] > 0
    clk = (d[ok, 0] / d[ok, 1] * 0.1)         # GHz
    loop_us = d[ok, 1] / 100.0
    med = float(clk.median())
    mfma_cycles = (4.0 * s * s * D * (34.0 / 32.0)) / (256 * 4 * 16384 / 16)     # per-SIMD matrix-pipe cycles of the launch (row sums included)
    print(f"{'zeros' if '--zeros' in sys.argv else 'random'} data: launch {ms:.3f} ms = {4.0 * s * s * D / ms / 1e9:.1f} TFLOP/s; in-kernel clock median "
          f"{med:.3f} GHz (p5 {float(clk.quantile(0.05)):.3f}, p95 {float(clk.quantile(0.95)):.3f}; {int(ok.sum())} workgroups, steady loop "
          f"median {float(loop_us.median()):.0f} us each); matrix pipe busy at that clock: {mfma_cycles / (ms * 1e-3 * med * 1e9) * 100:.1f} % "
          f"(bf16 dense peak at that clock: {256 * 4 * 16384 / 16 * med * 1e9 / 1e12:.0f} TFLOP/s)")


if __name__ == "__main__":
    build() if "--build" in sys.argv else run()
