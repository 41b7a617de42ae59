#!/usr/bin/env python3
"""Where does a workgroup of gemm_a4_kernel spend its time outside the K loop?  Diagnostic build (-DGF_GEMM_STAMP=1) that stamps
s_memtime at kernel entry / loop entry / loop exit / kernel exit of every workgroup.
  python3 tools/gemm_a4_stamps.py --build      (CPU container)
  python3 tools/gemm_a4_stamps.py              (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
LIB = os.path.join(ROOT, "build", "ab", "libstamp_a4.so")


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    src = [os.path.join(CSRC, f) for f in ("gf_gemm.hip", "gf_abi.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD", "-DGF_GEMM_STAMP=1",
                    f"-I{CSRC}/../include", f"-I{CSRC}", "-o", LIB] + src, check=True)
    print("built", LIB)


def main():
    import torch
    lib = ctypes.CDLL(LIB)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.gf_gemm_bf16.argtypes = [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp, i64, vp, vp]
    lib.gf_debug_set_gemm_buffer.argtypes = [vp]
    S, D, F = 32760, 5120, 13824
    st = torch.cuda.current_stream().cuda_stream
    for name, (n, k) in {"D->D": (D, D), "D->F": (F, D), "F->D": (D, F)}.items():
        x = torch.randn((S, k), device="cuda").to(torch.bfloat16)
        w = (torch.randn((n, k), device="cuda") / k ** 0.5).to(torch.bfloat16)
        out = torch.empty((S, n), device="cuda", dtype=torch.bfloat16)
        nwg = -(-S // 256) * (n // 256)
        dbg = torch.zeros((nwg, 4), dtype=torch.int64, device="cuda")
        lib.gf_debug_set_gemm_buffer(dbg.data_ptr())
        for _ in range(3):
            lib.gf_gemm_bf16(x.data_ptr(), k, w.data_ptr(), k, None, out.data_ptr(), n, S, n, k, 0, None, 0, None, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.gf_gemm_bf16(x.data_ptr(), k, w.data_ptr(), k, None, out.data_ptr(), n, S, n, k, 0, None, 0, None, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        t = dbg.cpu().double()
        pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        rounds = nwg / 256
        # s_memtime ticks at a constant rate; the XCDs' counters are not aligned, so only differences inside a workgroup are used
        # and the tick is calibrated by assuming the CUs are never idle: rounds x mean(workgroup time) = launch time
        tick_us = ms * 1e3 / (rounds * float((t[:, 3] - t[:, 0]).mean()))
        q = lambda a, f: float(a.quantile(f)) * tick_us
        print(f"{name}: launch {ms:.3f} ms, {nwg} workgroups = {rounds:.1f} rounds; raw ticks per workgroup {float((t[:, 3] - t[:, 0]).mean()):.0f} "
              f"(tick = {tick_us * 1e3:.2f} ns if the CUs are never idle)")
        print(f"   per workgroup (mean / p10 / p90, us): prologue {float(pro.mean()) * tick_us:6.2f} {q(pro, .1):6.2f} {q(pro, .9):6.2f} | "
              f"K loop {float(loop.mean()) * tick_us:7.2f} {q(loop, .1):7.2f} {q(loop, .9):7.2f} ({float(loop.mean()) * tick_us * 1e3 / (k / 64):.0f} ns per K tile) | "
              f"epilogue {float(epi.mean()) * tick_us:6.2f} {q(epi, .1):6.2f} {q(epi, .9):6.2f}")
        print(f"   raw ticks: prologue {float(pro.mean()):.1f}, loop {float(loop.mean()):.1f}, epilogue {float(epi.mean()):.1f}")


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
