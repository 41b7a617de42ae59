#!/usr/bin/env python3
"""A/B of flash-attention builds in ONE process (interleaved rounds, same device): each variant = extra -D flags.
  python3 tools/attn_ab.py --build name1:-DFLAG=1 name2:-DFLAG=0 ...     (CPU container: build/ab/libgf_<name>.so)
  python3 tools/attn_ab.py [--s 32760] [--rounds 4]                       (GPU: times every built variant, checks they agree)"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
OUT = os.path.join(ROOT, "build", "ab")


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    for f in glob.glob(os.path.join(OUT, "libgf_*.so")):
        os.remove(f)
    src = [os.path.join(CSRC, f) for f in ("gf_attention.hip", "gf_abi.hip")]
    for spec in specs:
        name, _, flags = spec.partition(":")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD",
                        "-mllvm", "-amdgpu-mfma-vgpr-form", f"-I{CSRC}/../include", f"-I{CSRC}",
                        "-o", os.path.join(OUT, f"libgf_{name}.so")] + [f for f in flags.split(",") if f] + src, check=True)
        print("built", name, flags, flush=True)


def run(s, rounds):
    import torch
    H, D = 40, 5120
    q, k, v = (torch.randn((s, D), device="cuda").to(torch.bfloat16) for _ in range(3))
    if "--zeros" in sys.argv:
        q, k, v = (torch.zeros_like(t) for t in (q, k, v))
    if "--qscale" in sys.argv:      # logits x f: near-one-hot softmax rows (f = 8 is bench.py --peaky 8), the rescale paths fire more often
        q = (q.float() * float(sys.argv[sys.argv.index("--qscale") + 1])).to(torch.bfloat16)
    kv_pad = -(-s // 64) * 64
    vt = torch.zeros((H * 128 * kv_pad,), dtype=torch.bfloat16, device="cuda")
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    st = torch.cuda.current_stream().cuda_stream
    flops = 4.0 * s * s * D
    libs, outs = {}, {}
    for path in sorted(glob.glob(os.path.join(OUT, "libgf_*.so"))):
        lib = ctypes.CDLL(path)
        for fn in ("gf_transpose_v", "gf_transpose_v32"):
            getattr(lib, fn).argtypes = [vp, i64, vp, i64, i64, i64, vp]
        for fn in ("gf_flash_attn_fwd_vt", "gf_flash_attn_fwd_vt32"):
            getattr(lib, fn).argtypes = [vp] * 5 + [i64] * 8 + [ctypes.c_float, vp]
        libs[os.path.basename(path)[6:-3]] = lib
    best = {n: 1e9 for n in libs}
    for rnd in range(rounds):
        for name, lib in libs.items():
            o = torch.empty_like(q)

            k2 = name.startswith("k2")      # variant names starting with "k2" time kernel 2, everything else kernel 3
            tr, fa = (lib.gf_transpose_v, lib.gf_flash_attn_fwd_vt) if k2 else (lib.gf_transpose_v32, lib.gf_flash_attn_fwd_vt32)

            def call():
                assert tr(v.data_ptr(), D, vt.data_ptr(), s, kv_pad, H, st) == 0
                assert fa(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), None, s, s, kv_pad, H, 128, D, D, D, 128 ** -0.5, st) == 0
            call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                call()
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / 6)
            outs[name] = o
    ref = next(iter(outs.values())).float()
    for name, ms in best.items():
        d = float((outs[name].float() - ref).norm() / ref.norm())
        print(f"{name:24s} {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s   rel diff vs first {d:.2e}", flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build([a for a in sys.argv[1:] if a != "--build"])
    else:
        s = int(sys.argv[sys.argv.index("--s") + 1]) if "--s" in sys.argv else 32760
        r = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
        run(s, r)
