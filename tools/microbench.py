#!/usr/bin/env python3
"""Kernel micro-benchmarks at the production shapes (random data, events on the launch stream).
    python3 tools/microbench.py attn [--iters 10]     self-attention S=32760, 40 heads, d=128
    python3 tools/microbench.py gemm [--iters 10]     the four GEMM shapes of a DiT block (+fp8 variants with --fp8)
    python3 tools/microbench.py rows                  LayerNorm+modulate / RMSNorm+RoPE HBM rates
Used for A/B-ing kernel changes in ONE process and as the target command of rocprofv3 PMC passes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from goal_force_amd import ops

S, D, H, F = 32760, 5120, 40, 13824
BF = torch.bfloat16


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["attn", "gemm", "rows", "cross", "attnbwd"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--s", type=int, default=S)
    ap.add_argument("--zeros", action="store_true", help="attn: all-zero q/k/v as well (same instruction stream, no data toggling: "
                    "how much of the time is the clock the chip grants an MFMA stream on REAL data)")
    ap.add_argument("--ramp", type=float, default=0.0, help="attn: also time an ADVERSARIAL input for the lazy-rescale softmax — every query's logits "
                    "rise linearly along the key index by this many log2 units per 64-key tile (> 6 forces a rescale of O in every tile; "
                    "no trained attention looks like this: the whole mass sits on the last keys) — and check it against fp64 on sampled rows")
    ap.add_argument("--ab", action="store_true", help="gemm: interleaved A/B of GF_GEMM_KERNEL=a4 (4 waves, shipped) and ph (8 waves)")
    ap.add_argument("--only", choices=["dd", "ffn1", "ffn2"], default=None, help="gemm: ONE block shape with the epilogue the block runs it "
                    "with (dd: bias; ffn1: GELU; ffn2: gate*+resid), bf16 — or e4m3 with --fp8: the target of the per-shape PMC passes "
                    "(tools/profile_r04.sh), so that a counter table holds one shape per kernel instantiation")
    ap.add_argument("--ref", action="store_true", help="also time torch F.linear (hipBLASLt) on the same shapes: a yardstick, not a product path")
    a = ap.parse_args()
    torch.manual_seed(0)
    s = a.s
    if a.what in ("attn", "cross"):
        skv = s if a.what == "attn" else 512
        q = torch.randn((s, D), device="cuda").to(BF)
        k = torch.randn((skv, D), device="cuda").to(BF)
        v = torch.randn((skv, D), device="cuda").to(BF)
        o = torch.empty_like(q)
        med, mn = timeit(lambda: ops.flash_attn(q, k, v, H, out=o), a.iters)
        fl = 4.0 * s * skv * D
        print(f"flash_attn S={s} Skv={skv} H={H}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms ({fl / mn / 1e9:.1f} TFLOP/s)")
        if a.ref:      # yardstick only: torch's F.scaled_dot_product_attention — the backend the reference falls back to on this
            # stack (DIT:55-60: no flash_attn / sageattention packages) — same box, same process, interleaved rounds
            import torch.nn.functional as TF
            hd = D // H
            q4, k4, v4 = (t.view(1, -1, H, hd).transpose(1, 2) for t in (q, k, v))       # the reference's 'b s (n d) -> b n s d'
            try:
                ref_o = TF.scaled_dot_product_attention(q4, k4, v4)
                torch.cuda.synchronize()
                err = float((ref_o.transpose(1, 2).reshape(s, D).float() - ops.flash_attn(q, k, v, H).float()).norm() / ref_o.float().norm())
                ours, theirs = [], []
                for _ in range(3):
                    ours.append(timeit(lambda: ops.flash_attn(q, k, v, H, out=o), a.iters)[0])
                    theirs.append(timeit(lambda: TF.scaled_dot_product_attention(q4, k4, v4), a.iters)[0])
                print(f"   yardstick F.scaled_dot_product_attention: medians {', '.join(f'{t:.3f}' for t in theirs)} ms -> "
                      f"{fl / min(theirs) / 1e9:.1f} TFLOP/s best;  ours interleaved: {', '.join(f'{t:.3f}' for t in ours)} ms -> "
                      f"{fl / min(ours) / 1e9:.1f} TFLOP/s best;  rel-L2 between the two outputs {err:.2e}")
            except Exception as e:      # noqa: BLE001 — a yardstick that cannot run is reported, not fatal
                print(f"   yardstick F.scaled_dot_product_attention failed: {type(e).__name__}: {str(e)[:200]}")
        if a.ramp > 0:
            import math
            hd = D // H
            tiles = (skv + 63) // 64
            top = a.ramp * tiles * math.log(2.0) * math.sqrt(hd)            # q.k of the LAST key so that the logit rises `ramp` log2 units per tile
            amp = math.sqrt(top)
            u = torch.ones((hd,), device="cuda") / math.sqrt(hd)
            qr = (amp * u).repeat(H)[None, :].expand(s, D).contiguous().to(BF)
            kr = ((torch.arange(skv, device="cuda", dtype=torch.float32) / skv)[:, None] * (amp * u).repeat(H)[None, :]).to(BF)
            vr = torch.randn((skv, D), device="cuda").to(BF)
            rounds = {"random": [], "ramp": []}
            for _ in range(3):
                rounds["random"].append(timeit(lambda: ops.flash_attn(q, k, v, H, out=o), a.iters)[0])
                rounds["ramp"].append(timeit(lambda: ops.flash_attn(qr, kr, vr, H, out=o), a.iters)[0])
            got = ops.flash_attn(qr, kr, vr, H)
            rows = torch.tensor([0, 1, s // 2, s - 1], device="cuda")
            h0 = 3
            sc = (qr[rows, h0 * hd:(h0 + 1) * hd].double() @ kr[:, h0 * hd:(h0 + 1) * hd].double().T) / math.sqrt(hd)
            ref = torch.softmax(sc, dim=-1) @ vr[:, h0 * hd:(h0 + 1) * hd].double()
            err = float((got[rows, h0 * hd:(h0 + 1) * hd].double() - ref).norm() / ref.norm())
            print(f"   ramp (+{a.ramp:g} log2 units per 64-key tile: {'every tile rescales' if a.ramp > 6 else 'below the rescale threshold of 6'}): medians "
                  f"{', '.join(f'{t:.3f}' for t in rounds['ramp'])} ms against random {', '.join(f'{t:.3f}' for t in rounds['random'])} ms "
                  f"(interleaved) -> x {min(rounds['ramp']) / min(rounds['random']):.3f};  rel-L2 vs fp64 on sampled rows {err:.2e}")
        if a.ab:       # interleaved rounds of kernel 3 (16x16x32 MFMA) and kernel 2 (32x32x16) in this process
            rounds = {"3": [], "2": []}
            for _ in range(3):
                for kern in ("3", "2"):
                    with ops.options(attn_k3=(kern == "3")):
                        rounds[kern].append(timeit(lambda: ops.flash_attn(q, k, v, H, out=o), a.iters)[0])
            for kern, ts in rounds.items():
                print(f"   A/B kernel {kern}: medians {', '.join(f'{t:.3f}' for t in ts)} ms -> {fl / min(ts) / 1e9:.1f} TFLOP/s best")
        if a.zeros:
            for name, mk in (("zeros", lambda t: torch.zeros_like(t)), ("ones", lambda t: torch.ones_like(t)),
                             ("random again", lambda t: t)):
                qq, kk, vv = mk(q), mk(k), mk(v)
                for kern in ("3", "2"):
                    with ops.options(attn_k3=(kern == "3")):
                        med, mn = timeit(lambda: ops.flash_attn(qq, kk, vv, H, out=o), a.iters)
                    print(f"   {name:13s} kernel {kern}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms")
    elif a.what == "attnbwd":
        q = torch.randn((s, D), device="cuda").to(BF)
        k = torch.randn((s, D), device="cuda").to(BF)
        v = torch.randn((s, D), device="cuda").to(BF)
        do = torch.randn((s, D), device="cuda").to(BF)
        o, lse = ops.flash_attn_lse(q, k, v, H)
        med, mn = timeit(lambda: ops.flash_attn_bwd(q, k, v, o, do, lse, H), a.iters)
        fl = 10.0 * s * s * D       # the 5 necessary products (executed: dQ 3 + dK/dV 4 = 7)
        ex = 1.4
        print(f"flash_attn_bwd S={s} H={H}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s algorithmic, "
              f"{ex * fl / med / 1e9:.1f} executed)")
    elif a.what == "gemm" and a.only:
        n, k = {"dd": (D, D), "ffn1": (F, D), "ffn2": (D, F)}[a.only]
        inp = torch.randn((s, k), device="cuda").to(BF)
        w = (torch.randn((n, k), device="cuda") / k ** 0.5).to(BF)
        b = torch.randn((n,), device="cuda").to(BF)
        out = torch.empty((s, n), device="cuda", dtype=BF)
        kw = {}
        if a.only == "ffn1":
            kw = dict(epilogue=ops.EPI_BIAS_GELU_TANH)
        if a.only == "ffn2":
            kw = dict(epilogue=ops.EPI_BIAS_GATE_RESID, resid=torch.randn((s, n), device="cuda").to(BF), gate=torch.randn((n,), device="cuda").to(BF))
        fl = 2.0 * s * n * k
        if a.fp8:
            w8 = ops.cast_fp8(w)
            x8, sc = ops.quant_fp8_rowscale(inp)
            med, mn = timeit(lambda: ops.gemm_fp8(x8, sc, w8, b, out=out, **kw), a.iters)
        else:
            med, mn = timeit(lambda: ops.gemm(inp, w, b, out=out, **kw), a.iters)
        print(f"gemm {'fp8' if a.fp8 else 'bf16'} {a.only} [{s},{k}]x[{n},{k}]^T: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms")
    elif a.what == "gemm":
        x = torch.randn((s, D), device="cuda").to(BF)
        xf = torch.randn((s, F), device="cuda").to(BF)
        for name, (inp, n, k) in {"D->D": (x, D, D), "D->F (ffn1)": (x, F, D), "F->D (ffn2)": (xf, D, F)}.items():
            w = (torch.randn((n, k), device="cuda") / k ** 0.5).to(BF)
            b = torch.randn((n,), device="cuda").to(BF)
            out = torch.empty((s, n), device="cuda", dtype=BF)
            fl = 2.0 * s * n * k
            med, mn = timeit(lambda: ops.gemm(inp, w, b, out=out), a.iters)
            print(f"gemm bf16 {name}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms")
            if a.ab:       # interleaved rounds of the 4-wave kernel and the 8-wave phased kernel in this process
                rounds = {"a4": [], "ph": []}
                for _ in range(3):
                    for kern in ("a4", "ph"):
                        with ops.options(prefer_8wave=int(kern == "ph")):
                            rounds[kern].append(timeit(lambda: ops.gemm(inp, w, b, out=out), a.iters)[0])
                for kern, ts in rounds.items():
                    print(f"   A/B {kern} {name}: medians {', '.join(f'{t:.3f}' for t in ts)} ms -> {fl / min(ts) / 1e9:.1f} TFLOP/s best")
            if n == D:
                res = torch.randn((s, n), device="cuda").to(BF)
                gate = torch.randn((n,), device="cuda").to(BF)
                med, mn = timeit(lambda: ops.gemm(inp, w, b, out=out, epilogue=ops.EPI_BIAS_GATE_RESID, resid=res, gate=gate), a.iters)
                print(f"gemm bf16 {name} + gate*y+resid epilogue: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms")
            if n == F:
                med, mn = timeit(lambda: ops.gemm(inp, w, b, out=out, epilogue=ops.EPI_BIAS_GELU_TANH), a.iters)
                print(f"gemm bf16 {name} + GELU(tanh) epilogue: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms")
            if a.ref:
                med, mn = timeit(lambda: torch.nn.functional.linear(inp, w, b), a.iters)
                print(f"   torch F.linear {name}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms")
            if a.fp8:
                w8 = ops.cast_fp8(w)
                x8, sc = ops.quant_fp8_rowscale(inp)
                med, mn = timeit(lambda: ops.gemm_fp8(x8, sc, w8, b, out=out), a.iters)
                medq, _ = timeit(lambda: ops.quant_fp8_rowscale(inp), a.iters)
                print(f"gemm fp8  {name}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), quantise {medq:.3f} ms")
    else:
        x = torch.randn((s, D), device="cuda").to(BF)
        sc = torch.randn((D,), device="cuda").to(BF)
        out = torch.empty_like(x)
        med, _ = timeit(lambda: ops.layernorm_modulate(x, scale1p=sc, shift=sc, out=out), a.iters)
        print(f"layernorm_modulate: {med:.3f} ms = {2 * x.numel() * 2 / med / 1e9:.2f} TB/s")
        cos = torch.randn((s, 64), device="cuda")
        med, _ = timeit(lambda: ops.rmsnorm_rope(x, sc, cos, cos), a.iters)
        print(f"rmsnorm_rope: {med:.3f} ms = {(2 * x.numel() * 2 + 2 * cos.numel() * 4) / med / 1e9:.2f} TB/s")


if __name__ == "__main__":
    main()
