#!/usr/bin/env python3
"""Build-time check of the accumulator hand-off in gemm_a4_kernel (gf_gemm.hip) and conv_a4_kernel (gf_conv_a4.hip).

The K loop is one asm statement that leaves the 256 fp32 accumulators in a[0:255]; the epilogue fetches them with separate
`v_accvgpr_read_b32` asm statements.  Between the two the compiler sees the AGPRs as free, so a future hipcc could legally park
a spill or a load result in one of them and silently corrupt an accumulator.  This script compiles gf_gemm.hip to gfx950
assembly the Makefile keeps from the compile that produced the shipped object (`-save-temps=obj` on gf_gemm.o: same flags, same
compiler run — build/csrc/gf_gemm-hip-amdgcn-amd-amdhsa-gfx950.s; only when that file is missing or older than the source is the
file compiled here, with the Makefile's flags and the compiler's stderr passed through) and verifies for EVERY gemm_a4_kernel instantiation that between the end of the loop's asm block and the last
accumulator read the ONLY instructions that mention an AGPR are those reads, each register read exactly once.

    python tools/check_a4_agpr.py            # exit code 0 = safe; called by __graft_entry__.build() and tests/test_host_cpu.py
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "goal_force_amd", "csrc", "gf_gemm.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
AGPR = re.compile(r"\ba\d+\b|\ba\[(?:0x[0-9a-f]+|\d+)(?::\d+)?\]")


SAVED = os.path.join(ROOT, "build", "csrc", "gf_gemm-hip-amdgcn-amd-amdhsa-gfx950.s")
# (source, saved assembly, files it depends on, kernel name, the accumulators its loop leaves behind)
UNITS = [
    (SRC, SAVED, ("gf_gemm_a4_loop.inc", "gf_gemm_a4f8_loop.inc", "gf_common.h"), "gemm_a4_kernel", set(range(256))),
    (os.path.join(ROOT, "goal_force_amd", "csrc", "gf_conv_a4.hip"), os.path.join(ROOT, "build", "csrc", "gf_conv_a4-hip-amdgcn-amd-amdhsa-gfx950.s"),
     ("gf_conv_a4_loop.inc", "gf_common.h"), "conv_a4_kernel", {(i * 8 + j) * 4 + r for i in range(8) for j in range(6) for r in range(4)}),
]


def device_asm(src=SRC, saved=SAVED, dep_names=("gf_gemm_a4_loop.inc", "gf_gemm_a4f8_loop.inc", "gf_common.h")):
    deps = [src] + [os.path.join(ROOT, "goal_force_amd", "csrc", f) for f in dep_names]
    if os.path.exists(saved) and all(os.path.getmtime(saved) >= os.path.getmtime(d) for d in deps):
        return open(saved).read(), saved
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "unit.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/goal_force_amd/csrc",
               "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-DGF_BUILD", "--cuda-device-only", "-S", "-o", out, src]
        subprocess.run(cmd, check=True)             # the Makefile's CXXFLAGS; compiler diagnostics go to our stderr
        return open(out).read(), f"a fresh compile of {os.path.basename(src)} (no saved assembly of the shipped object)"


def check(text, kernel="gemm_a4_kernel", want=None):
    """Returns {kernel: number of accumulator reads}; raises AssertionError on a violation."""
    want = set(range(256)) if want is None else want
    res = {}
    kernels = re.findall(r"^(_ZN\S*" + kernel + r"\S*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
    assert kernels, f"no {kernel} instantiation found in the device assembly"
    for name, body in kernels:
        lines = [l.split(";")[0].strip() for l in body.splitlines()]
        lines = [l for l in lines if l and not l.startswith((".", "//"))]
        # the loop's asm block ends with: s_waitcnt vmcnt(0) / s_nop 7 ... / s_mov_b32 m0, s52 / s_barrier
        ends = [i for i, l in enumerate(lines) if re.match(r"s_mov_b32 m0, s52$", l)]
        assert len(ends) >= 1, f"{name}: end of the K loop's asm block not found"
        start = ends[-1] + 1
        reads = [i for i, l in enumerate(lines) if l.startswith("v_accvgpr_read_b32") and i > start]
        assert reads, f"{name}: no accumulator reads behind the loop"
        seen = set()
        for i in range(start, reads[-1] + 1):
            l = lines[i]
            if not AGPR.search(l):
                continue
            m = re.match(r"v_accvgpr_read_b32 v\d+, a\[?(0x[0-9a-f]+|\d+)\]?$", l)
            assert m, f"{name}: `{l}` touches an AGPR between the K loop and the last accumulator read"
            r = int(m.group(1), 0)
            assert r not in seen, f"{name}: a{r} is read twice"
            seen.add(r)
        assert seen == want, f"{name}: {len(seen)} of {len(want)} accumulators are read"
        # and nothing before the loop may leave a value in an AGPR that the loop does not overwrite: the loop zeroes all 256
        res[name] = len(seen)
    return res


def main():
    for src, saved, deps, kernel, want in UNITS:
        text, where = device_asm(src, saved, deps)
        try:
            res = check(text, kernel, want)
            # the convolution kernel must not spill at all (its epilogue is small; the fp8 GEMM's residual epilogue spills 28 bytes
            # OUTSIDE the loop, which the AGPR check above covers)
            for m in [] if kernel != "conv_a4_kernel" else re.finditer(r"^\s*\.amdhsa_kernel (\S*" + kernel + r"\S*).*?\.amdhsa_private_segment_fixed_size (\d+)", text, re.S | re.M):
                assert int(m.group(2)) == 0, f"{m.group(1)}: {m.group(2)} bytes of scratch (a spill next to a hand-allocated register plan)"
        except AssertionError as e:
            print(f"check_a4_agpr: FAILED on {where}: {e}", file=sys.stderr)
            return 1
        print(f"checked {where}")
        for k, n in res.items():
            print(f"ok  {k}: {n} accumulator reads, no other AGPR access between the loop and the last read")
    return 0


if __name__ == "__main__":
    sys.exit(main())
