#!/usr/bin/env python3
"""Builds GEMM library variants for tools/gemm_ab.py: build/ab/libgemm_<name>.so, each from a freshly generated K loop.
  python3 tools/gemm_variants.py name1:A4_WAIT_SLOT=100,A4_B_FIRST=1 name2::-DGF_GROUP_M=4 base
(spec = name[:generator env k=v,...[:extra hipcc flags,...]])"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
OUT = os.path.join(ROOT, "build", "ab")


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in glob.glob(os.path.join(OUT, "libgemm_*.so")):
        os.remove(f)
    src = [os.path.join(CSRC, f) for f in ("gf_gemm.hip", "gf_abi.hip")]
    for spec in sys.argv[1:]:
        name, _, rest = spec.partition(":")
        envs, _, flags = rest.partition(":")
        inc = os.path.join(OUT, f"a4_loop_{name}.inc")
        env = dict(os.environ, A4_OUT=inc)
        env.update(kv.split("=", 1) for kv in envs.split(",") if kv)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gemm_a4.py")], env=env, check=True, stdout=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD",
                        f"-I{CSRC}/../include", f"-I{CSRC}", f'-DGF_A4_LOOP_INC="{inc}"',
                        "-o", os.path.join(OUT, f"libgemm_{name}.so")] + [f for f in flags.split(",") if f] + src, check=True)
        print("built", name, envs, flags, flush=True)


if __name__ == "__main__":
    main()
