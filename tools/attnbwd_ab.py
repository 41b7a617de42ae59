#!/usr/bin/env python3
"""A/B of flash-attention BACKWARD variants in ONE process (interleaved rounds, S = 32760, 40 heads):
    python tools/attnbwd_ab.py --build     (CPU) builds build/ab/libbwd_<name>.so for every variant in VARIANTS
    python tools/attnbwd_ab.py             (GPU) times them, checks each against the first (rel-L2 of dQ / dK / dV)
A variant = -D flags for gf_attention_bwd.hip (GF_BWD_SEED, GF_BWD_HALVES, ...); the forward (q, k, v -> o, lse) comes from the
shipped library."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the variants these builds select are not in the product sources: tools/experimental_tree.sh re-creates them (tools/patches/)
CSRC = os.environ.get("GF_CSRC", os.path.join(ROOT, "build", "experimental", "csrc"))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "ab")
VARIANTS = {      # KV16_WHATIF bits (32-key kernel only; timing only, wrong results): 1 no DMA in the loop, 4 no P hand-off, 8 no barrier, 16 no counted wait, 32 L2-hot DMA
    "base": [],
    "qscale": ["-DGF_BWD_QSCALE=1"],        # round 4's experiment: P rebuilt from the forward's pre-scaled Q' (no fma per score)
}
for spec in os.environ.get("BWD_AB_EXTRA", "").split(";"):     # name:flag,flag
    if spec:
        n, _, fl = spec.partition(":")
        VARIANTS[n] = fl.split(",")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = [os.path.join(CSRC, f) for f in ("gf_attention_bwd.hip", "gf_abi.hip")]
    for old in os.listdir(OUT):
        if old.startswith("libbwd_"):
            os.remove(os.path.join(OUT, old))
    for name, flags in VARIANTS.items():
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGF_BUILD", "-fno-slp-vectorize",
                        f"-I{CSRC}/../include", f"-I{CSRC}", "-o", os.path.join(OUT, f"libbwd_{name}.so")] + flags + src,
                       check=True)
        print("built", name, flags, flush=True)


def run():
    import torch
    from goal_force_amd import ops
    S, H, D = int(os.environ.get("BWD_AB_S", "32760")), 40, 5120
    torch.manual_seed(0)
    q, k, v, do = (torch.randn((S, D), device="cuda").to(torch.bfloat16) for _ in range(4))
    o, lse = ops.flash_attn_lse(q, k, v, H)
    from goal_force_amd import _lib
    ws = torch.empty((int(_lib.load().gf_flash_attn_bwd_workspace_bytes(S, S, H)),), dtype=torch.uint8, device="cuda")
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    st = torch.cuda.current_stream().cuda_stream
    libs, outs, best = {}, {}, {}
    import glob
    for path in sorted(glob.glob(os.path.join(OUT, "libbwd_*.so"))):      # every built variant (incl. hand-built ones, e.g. an older source)
        lib = ctypes.CDLL(path)
        lib.gf_flash_attn_bwd.argtypes = [vp] * 10 + [i64] * 12 + [ctypes.c_float, vp]
        libs[os.path.basename(path)[7:-3]] = lib
    if os.environ.get("BWD_AB_V1", "1") == "1":      # the shipped library's first kernels (GF_ATTN_BWD=v1) as the yardstick
        for lib in libs.values():
            lib.gf_reload_options()                 # every library reads the knobs for itself: the variants without GF_ATTN_BWD
        os.environ["GF_ATTN_BWD"] = "v1"
        v1 = ctypes.CDLL(os.path.join(ROOT, "goal_force_amd", "libgoalforce_hip.so"))     # the handle ops uses (the forward is not affected)
        v1.gf_flash_attn_bwd.argtypes = [vp] * 10 + [i64] * 12 + [ctypes.c_float, vp]
        v1.gf_reload_options()
        del os.environ["GF_ATTN_BWD"]
        libs = {"shipped_v1": v1, **libs}
    bufs = [torch.empty_like(q) for _ in range(3)]

    def call(lib):
        rc = lib.gf_flash_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), ws.data_ptr(),
                                   bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), S, S, H, 128, D, D, D, D, D, D, D, D,
                                   128 ** -0.5, st)
        assert rc == 0
    for rnd in range(int(os.environ.get("BWD_AB_ROUNDS", "3"))):
        for name, lib in libs.items():
            call(lib)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                call(lib)
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) / 3)
            if rnd == 0:
                outs[name] = [b.float().clone() if S <= 8192 else b[:2048].float().clone() for b in bufs]
    first = next(iter(outs.values()))
    fl = 10.0 * S * S * D
    for name, ms in best.items():
        err = max(float((a - b).norm() / b.norm()) for a, b in zip(outs[name], first))
        print(f"{name:18s} {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s algorithmic   max rel-L2 vs first {err:.2e}", flush=True)


if __name__ == "__main__":
    build() if "--build" in sys.argv else run()
