#!/usr/bin/env python3
"""What bounds the K loop of gemm_a4_kernel: timing-only variants of the asm loop (tools/gen_gemm_a4.py `whatif`; results
are wrong by construction, nothing here ships).
  python3 tools/gemm_a4_whatif.py --build       (CPU container: cross-compiles build/whatif/libgf_a4w.so with -DGF_A4_WHATIF)
  python3 tools/gemm_a4_whatif.py               (GPU: times the variants on the three DiT GEMM shapes, interleaved)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
LIB = os.path.join(ROOT, "build", "whatif", "libgf_a4w.so")
NAMES = {0: "shipped loop", 1: "no barriers / vmcnt waits", 2: "K position frozen (all staging hits L2)", 4: "no staging instructions",
         5: "no staging, no barriers", 16: "source chunks not permuted (linear 128-byte rows per 8 lanes)", 32: "no epilogue", 64: "no vmcnt wait (barriers kept)", 128: "no barriers (vmcnt wait kept)"}


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = [os.path.join(CSRC, f) for f in ("gf_gemm.hip", "gf_abi.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD",
                    "-DGF_A4_WHATIF", f"-I{CSRC}/../include", f"-I{CSRC}", "-o", LIB] + src, check=True)
    print("built", LIB)


def run():
    import ctypes
    import torch
    lib = ctypes.CDLL(LIB)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.gf_gemm_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp, i64, vp, vp]
    S, D, F = 32760, 5120, 13824
    st = torch.cuda.current_stream().cuda_stream
    for name, (n, k) in {"D->D": (D, D), "D->F": (F, D), "F->D": (D, F)}.items():
        x = torch.randn((S, k), device="cuda").to(torch.bfloat16)
        w = (torch.randn((n, k), device="cuda") / k ** 0.5).to(torch.bfloat16)
        out = torch.empty((S, n), device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * S * n * k

        def call():
            assert lib.gf_gemm_bf16(x.data_ptr(), k, w.data_ptr(), k, None, out.data_ptr(), n, S, n, k, 0, None, 0, None, st) == 0
        best = {}
        for rnd in range(3):
            for m in NAMES:
                os.environ["GF_A4_WHATIF"] = str(m)
                lib.gf_reload_options()    # the library reads its knobs once per process
                for _ in range(2):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best[m] = min(best.get(m, 1e9), e0.elapsed_time(e1) / 6)
        for m, ms in best.items():
            print(f"{name}  whatif {m}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s-equivalent   {NAMES[m]}", flush=True)


if __name__ == "__main__":
    build() if "--build" in sys.argv else run()
