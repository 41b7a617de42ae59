#!/usr/bin/env python3
"""Full-depth parity of the HIP sampling path against the fp32-math oracle ON THE GPU — test infrastructure.

    python tests/fullsize_parity.py [--layers 40 --cn-layers 10 --grid 21 30 52 --steps 4] [--fp8] [--out gpurun_out/x.json]

What the reference computes per video is 100 x `model_fn_wan_video` (40 DiT + 10 ControlNet blocks,
src/goal_force/wan_video_new.py:1503-1570) inside the CFG / Euler loop (wan_video_new.py:697-723).  The goldens under
tests/golden compare at most 2 + 1 blocks at 72 tokens; this harness runs the WHOLE stack at the production size:

  (a) one high-noise forward (step 0, cond branch): rel-L2 of the residual stream after chosen DiT blocks and of the noise
      prediction — HIP vs fp32 math, next to the reference's own bf16 arithmetic vs fp32 math (the "noise floor": what two
      correct bf16 implementations may differ by);
  (b) a K-step CFG loop (K-step shift-5 schedule => the expert switch is inside for K >= 2): latents per step for the same
      three trajectories, and the PSNR of the tiled-decoded uint8 frames of the final latents;
  (c) with --fp8 the same for BASELINE config 5: HIP fp8 kernels vs the oracle graph with every block Linear replaced by a
      LIVE torch._scaled_mm through the call sequence of diffsynth/vram_management/layers.py:115-151.

The three arithmetic modes share weights (random-init bf16, bench.py's seeds), inputs and the bf16-rounded timestep
(wan_video_new.py:707):
  fp32  = oracle.wan_oracle graph, fp32 tensors, attention in query chunks (never materialises S x S x heads);
  bf16  = the same oracle graph on bf16 tensors through torch's own ROCm kernels (F.linear, F.scaled_dot_product_attention,
          ...) — the arithmetic the reference itself runs on this stack;
  hip   = goal_force_amd (the product).
The oracle is the checker here, never the thing measured; nothing under goal_force_amd/ imports this file.
tests/test_fulldepth_gpu.py runs a reduced configuration of the same code as a gated test.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections.abc import Mapping

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


class LazySD(Mapping):
    """State dict view of a module that converts a tensor to `dtype` when it is read (a 14 B-parameter expert never exists
    twice in fp32).  2-D weights of the DiT / ControlNet blocks carry `_gf_block_linear = True` so that the config-5 chain can
    route exactly those Linears through torch._scaled_mm (VRAM:113: only the wrapped block Linears compute in fp8)."""

    def __init__(self, module, dtype):
        self.sd, self.dtype = dict(module.state_dict()), dtype

    def __getitem__(self, k):
        p = self.sd[k]
        t = p.to(self.dtype)
        if "blocks." in k and k.endswith(".weight") and p.dim() == 2 and "norm" not in k:
            t = t.clone() if t is p else t
            t._gf_block_linear = True
        return t

    def __iter__(self):
        return iter(self.sd)

    def __len__(self):
        return len(self.sd)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def psnr_u8(a, b):
    mse = float((a.float() - b.float()).pow(2).mean())
    return float("inf") if mse == 0 else 10.0 * torch.log10(torch.tensor(255.0 ** 2 / mse)).item()


PRETRAINED = None      # {"models_dir": ..., "ckpt": ...} when main() found trained weights to load (--models-dir / --model-ckpt-path)
PEAKY = 1.0            # --peaky f: every self-attention's norm_q weight x f (logits x f: near-one-hot softmax rows, what trained attention looks like)


def make_peaky(pipe):
    """Random-init attention is near-uniform — the friendliest data for a flash-attention kernel and for its bf16 roundings.  Scaling the
    self-attention's norm_q weights scales its logits: the same model with peaky softmax rows (bench.py's `data_sensitivity` times this;
    here its PARITY is measured).  The weights live in the modules, so the oracle runs (LazySD) see the same scaled weights."""
    if PEAKY == 1.0:
        return
    for m in (pipe.dit, pipe.dit2, pipe.controlnet, pipe.controlnet2):
        if m is None:
            continue
        for blk in m.modules():
            if hasattr(blk, "self_attn") and hasattr(blk.self_attn, "norm_q"):
                blk.self_attn.norm_q.weight.data.mul_(PEAKY)


def pretrained_paths(models_dir):
    """The files the reference launcher loads (INF:81-106), or None when any expert shard or the VAE is missing under models_dir."""
    hi = [os.path.join(models_dir, "Wan2.2-I2V-A14B", "high_noise_model", f"diffusion_pytorch_model-0000{i}-of-00006.safetensors") for i in range(1, 7)]
    lo = [p.replace("high_noise_model", "low_noise_model") for p in hi]
    vae = os.path.join(models_dir, "Wan2.1-T2V-1.3B", "Wan2.1_VAE.pth")
    return (hi, lo, vae) if all(os.path.exists(p) for p in hi + lo + [vae]) else None


def build_pretrained(models_dir, ckpt, cn_layers, dev):
    """TRAINED weights through the product's own `from_pretrained` (the reference's call, INF:81-111) — the run the build image cannot
    make (no checkpoints, no network): `python tests/fullsize_parity.py --steps 50 --models-dir ./models/Wan-AI --model-ckpt-path
    checkpoints/.../step-N.safetensors`.  The text encoder is not loaded: the conditioning tensors stay the seeded synthetic ones."""
    from goal_force_amd.pipeline import ModelConfig, WanVideoPipeline
    hi, lo, vae = pretrained_paths(models_dir)
    pipe = WanVideoPipeline.from_pretrained(torch_dtype=torch.bfloat16, device=dev, controlnet=True, controlnet_num_layers=cn_layers,
                                            model_configs=[ModelConfig(path=hi), ModelConfig(path=lo), ModelConfig(path=vae)])
    if ckpt:
        pipe.load_controlnet_weights(pipe.controlnet, ckpt, torch_dtype=torch.bfloat16)        # INF:108
    return config_of(pipe.dit), pipe


def config_of(dit):
    """The oracle's cfg dict of a built WanModel."""
    return dict(dim=dit.dim, in_dim=dit.in_dim, ffn_dim=dit.blocks[0].ffn_dim, out_dim=dit.out_dim, text_dim=dit.text_embedding[0].in_features,
                freq_dim=dit.freq_dim, eps=dit.eps, patch_size=dit.patch_size, num_heads=dit.num_heads, num_layers=len(dit.blocks))


def build(layers, cn_layers, dev, need_low=True):
    if PRETRAINED is not None:
        return build_pretrained(PRETRAINED["models_dir"], PRETRAINED["ckpt"], cn_layers, dev)
    from goal_force_amd.dit import A14B_CONFIG
    from goal_force_amd.pipeline import WanVideoPipeline, build_random_controlnet, build_random_expert
    from goal_force_amd.vae import WanVideoVAE
    cfg = dict(A14B_CONFIG)
    cfg["num_layers"] = layers
    dit = build_random_expert(cfg, seed=100, device=dev)
    cn = build_random_controlnet(cn_layers, cfg, seed=300, device=dev)
    dit2 = build_random_expert(cfg, seed=200, device=dev) if need_low else None
    cn2 = build_random_controlnet(cn_layers, cfg, seed=400, device=dev, zero_convs_zero=True) if need_low else None
    torch.manual_seed(7)
    vae = WanVideoVAE().to(torch.bfloat16).to(dev)
    return cfg, WanVideoPipeline.from_modules(dit, dit2, cn, cn2, vae=vae, device=dev)


def inputs(pipe, grid, dev, sample=0):
    """bench.py's synthetic conditioning at the given latent grid (f, 2h, 2w)."""
    f, hh, ww = grid
    g = torch.Generator().manual_seed(1000 + sample)
    lat = pipe.generate_noise((1, 16, f, hh, ww), seed=sample)
    y = torch.randn((1, 20, f, hh, ww), generator=g)
    y[:, :4] = 0
    y[:, :4, 0] = 1
    control = torch.randn((1, 16, f, hh, ww), generator=g)
    td = pipe.dit.text_embedding[0].in_features               # 4096 at A14B size
    ctx_p, ctx_n = torch.randn((1, 512, td), generator=g), torch.randn((1, 512, td), generator=g)
    ctx_p[:, 40:] = 0
    ctx_n[:, 40:] = 0
    bf = torch.bfloat16
    return dict(latents=lat, y=y.to(bf).to(dev), control=control.to(bf).to(dev), ctx_p=ctx_p.to(bf).to(dev), ctx_n=ctx_n.to(bf).to(dev))


class OracleRunner:
    """The oracle's model_fn / CFG / Euler (oracle.wan_oracle: GF:1349-1591, 716, FM:72-82) on device tensors in one dtype."""

    def __init__(self, pipe, cfg, dtype, q_chunk=2048, fp8_chain=False):
        from oracle import wan_oracle as wo
        self.wo, self.cfg, self.dtype, self.q_chunk, self.fp8_chain = wo, cfg, dtype, q_chunk, fp8_chain
        self.experts = [(LazySD(pipe.dit, dtype), LazySD(pipe.controlnet, dtype), pipe.controlnet.num_layers)]
        if pipe.dit2 is not None:
            # the never-trained ControlNet2 (zero-convs exactly zero, GF:565): conv1d with zero weight and bias adds exact zeros,
            # x + 0 == x bitwise in every dtype, so the oracle skips its 10 blocks as the product does (saves GPU minutes only)
            z = pipe.controlnet2.all_zero()
            self.experts.append((LazySD(pipe.dit2, dtype), None if z else LazySD(pipe.controlnet2, dtype),
                                 0 if z else pipe.controlnet2.num_layers))

    def _linear(self):
        if not self.fp8_chain:
            return F.linear
        from make_fp8_golden_gpu import scaled_mm_linear      # VRAM:115-151 restated around the live torch._scaled_mm

        def lin(x, w, b=None):
            if getattr(w, "_gf_block_linear", False):
                return scaled_mm_linear(x, w, b)[0]
            return F.linear(x, w, b)
        return lin

    def forward(self, which, latents, ts_bf16, ctx, inp, tap=None):
        wo = self.wo
        dsd, csd, ncn = self.experts[which]
        old = (wo.LINEAR, wo.ATTENTION_Q_CHUNK)
        wo.LINEAR, wo.ATTENTION_Q_CHUNK = self._linear(), self.q_chunk
        try:
            dt = self.dtype
            return wo.model_fn(dsd, self.cfg, latents.to(dt), ts_bf16.to(dt), ctx.to(dt), inp["y"].to(dt), csd,
                               None if csd is None else inp["control"].to(dt), ncn, tap=tap)
        finally:
            wo.LINEAR, wo.ATTENTION_Q_CHUNK = old

    def loop(self, inp, n_steps, cfg_scale=5.0, boundary=0.875, tap0=None, log=None):
        """GF:697-723 — returns the latents after every step (list of n_steps tensors, dtype of the run) and the first
        forward's noise prediction."""
        wo = self.wo
        sigmas, timesteps = wo.flow_match_sigmas(n_steps, 5.0)
        lat = inp["latents"].to(self.dtype)
        out, first, cur = [], None, 0
        for i, ts in enumerate(timesteps):
            if ts.item() < boundary * 1000 and cur == 0 and len(self.experts) > 1:
                cur = 1
            tsb = ts.unsqueeze(0).to(torch.bfloat16).to(lat.device)          # GF:707: the timestep reaches model_fn bf16-rounded
            t0 = time.time()
            posi = self.forward(cur, lat, tsb, inp["ctx_p"], inp, tap=tap0 if i == 0 else None)
            if first is None:
                first = posi
            nega = self.forward(cur, lat, tsb, inp["ctx_n"], inp)
            lat = wo.euler_step(wo.cfg_combine(posi, nega, cfg_scale), i, lat, sigmas)
            torch.cuda.synchronize()
            if log:
                log(f"    oracle[{self.name()}] step {i} (expert {cur}): {time.time() - t0:.1f} s")
            out.append(lat)
        return out, first

    def name(self):
        return ("fp8-chain" if self.fp8_chain else "bf16") if self.dtype == torch.bfloat16 else "fp32"


def hip_loop(pipe, inp, n_steps, taps=None):
    """The product's denoise() one step at a time (public API: step_ids) so that every step's latents can be kept; the first
    step's cond forward is additionally run on its own with forward hooks on the chosen DiT blocks (same kernels, same inputs)."""
    lat = inp["latents"]
    first, tapped = None, {}
    if taps is not None:
        pipe.scheduler.set_timesteps(n_steps, shift=5.0)
        ts = pipe.scheduler.timesteps[0].unsqueeze(0).to(dtype=torch.bfloat16, device=lat.device)
        first, tapped = hip_forward(pipe, inp, ts, taps)
    out = []
    for i in range(n_steps):
        lat = pipe.denoise(lat, inp["ctx_p"], inp["ctx_n"], inp["y"], inp["control"], num_inference_steps=n_steps, cfg_scale=5.0,
                           controlnet=True, step_ids=[i])
        out.append(lat)
    torch.cuda.synchronize()
    return out, first, tapped


def hip_forward(pipe, inp, ts, taps):
    """One cond forward of the product (high-noise expert + ControlNet) with the residual stream kept after the DiT blocks `taps`."""
    tapped = {}
    hooks = [pipe.dit.blocks[i].register_forward_hook(lambda m, a, o, i=i: tapped.__setitem__(i, o.detach().clone().reshape(1, -1, o.shape[-1])))
             for i in taps if i < len(pipe.dit.blocks)]
    try:
        first = pipe.model_fn(dit=pipe.dit, controlnet=pipe.controlnet, latents=inp["latents"], timestep=ts, context=inp["ctx_p"],
                              y=inp["y"], control_signal_video_latents=inp["control"])
    finally:
        for h in hooks:
            h.remove()
    torch.cuda.synchronize()
    return first, tapped


def run_forward(layers=40, cn_layers=10, grid=(21, 60, 104), fp8=True, taps=(0, 9, 19, 39), q_chunk=2048, log=print, out_path=None):
    """ONE noise prediction at production size (GF:1503-1570: step 0 of the 50-step schedule, cond branch, high-noise expert with its
    ControlNet): the product against fp32 math next to the reference's bf16 arithmetic, at the residual stream after the DiT blocks
    `taps` and at the noise prediction; with `fp8` also the config-5 stack against the live torch._scaled_mm chain.  This is the
    part of run() that fits a gated test at 40 + 10 blocks and S = 32760 (one fp32 forward instead of 2 x steps of them)."""
    from goal_force_amd.dit import enable_fp8
    torch.set_grad_enabled(False)
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda", torch.cuda.current_device())
    taps = tuple(t for t in taps if t < layers)
    t_all = time.time()
    cfg, pipe = build(layers, cn_layers, dev, need_low=False)
    make_peaky(pipe)
    inp = inputs(pipe, grid, dev)
    tokens = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    pipe.scheduler.set_timesteps(50, shift=5.0)
    ts = pipe.scheduler.timesteps[0].unsqueeze(0).to(dtype=torch.bfloat16, device=dev)          # GF:707
    rep = {"config": {"layers": layers, "controlnet_layers": cn_layers, "latent": [1, 16, *grid], "tokens": tokens,
                      "what": "one cond forward, step 0 of the 50-step shift-5 schedule", "timestep_bf16": float(ts.float()),
                      "taps_after_dit_block": [t + 1 for t in taps], "weights": "random-init bf16 (bench.py seeds 100/300)" + ("" if PEAKY == 1.0 else f", self-attention logits x {PEAKY:g}"),
                      "device": torch.cuda.get_device_name(0), "fp32_attention_q_chunk": q_chunk}}
    log(f"  built {layers} + {cn_layers} blocks in {time.time() - t_all:.1f} s")
    t0 = time.time()
    hip_first, hip_tap = hip_forward(pipe, inp, ts, taps)
    log(f"  hip forward: {time.time() - t0:.1f} s")

    def oracle(dtype, chain=False):
        tap = {}
        o = OracleRunner(pipe, cfg, dtype, q_chunk, fp8_chain=chain)
        t0 = time.time()
        first = o.forward(0, inp["latents"].to(dtype), ts, inp["ctx_p"], inp, tap=lambda i, x: tap.__setitem__(i, x.clone()) if i in taps else None)
        torch.cuda.synchronize()
        log(f"  oracle[{o.name()}] forward: {time.time() - t0:.1f} s")
        return first, tap

    f32_first, f32_tap = oracle(torch.float32)
    b16_first, b16_tap = oracle(torch.bfloat16)

    def rows(name, tap, first):
        r = {"after_block": {str(i + 1): rel_l2(tap[i].float(), f32_tap[i]) for i in taps if i in tap},
             "noise_pred_step0_cond": rel_l2(first.float(), f32_first)}
        log(f"  {name} vs fp32: blocks {r['after_block']}  noise_pred {r['noise_pred_step0_cond']:.3e}")
        return r

    rep["hip_bf16_vs_fp32"] = rows("hip-bf16", hip_tap, hip_first)
    rep["ref_bf16_vs_fp32"] = rows("ref-bf16", b16_tap, b16_first)
    rep["hip_bf16_vs_ref_bf16"] = {"noise_pred_step0_cond": rel_l2(hip_first.float(), b16_first.float())}
    _dump(rep, out_path)
    if fp8:
        del b16_tap, b16_first
        for m in (pipe.dit, pipe.controlnet):
            enable_fp8(m)
        h8_first, h8_tap = hip_forward(pipe, inp, ts, taps)
        for m in (pipe.dit, pipe.controlnet):
            enable_fp8(m, False)
        c8_first, c8_tap = oracle(torch.bfloat16, chain=True)
        rep["hip_fp8_vs_fp32"] = rows("hip-fp8", h8_tap, h8_first)
        rep["scaled_mm_chain_vs_fp32"] = rows("scaled_mm-chain", c8_tap, c8_first)
        rep["hip_fp8_vs_scaled_mm_chain"] = {"noise_pred_step0_cond": rel_l2(h8_first.float(), c8_first.float())}
    rep["wall_s"] = time.time() - t_all
    _dump(rep, out_path)
    return rep


def decode_u8(pipe, lat):
    frames = pipe.vae.decode(lat.to(torch.bfloat16), tiled=True, tile_size=(30, 52), tile_stride=(15, 26))
    return pipe.frames_uint8(frames)


def _dump(rep, out_path):
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(rep, f, indent=1)


def run(layers=40, cn_layers=10, grid=(21, 60, 104), steps=4, fp8=False, taps=(0, 9, 19, 39), q_chunk=2048, decode=True, log=print,
        out_path=None):
    """Returns the report dict (see the module docstring).  grid = latent (f, H/8, W/8)."""
    from goal_force_amd.dit import enable_fp8
    torch.set_grad_enabled(False)
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda", torch.cuda.current_device())
    taps = tuple(t for t in taps if t < layers)
    cfg, pipe = build(layers, cn_layers, dev, need_low=steps >= 2)
    make_peaky(pipe)
    inp = inputs(pipe, grid, dev)
    tokens = grid[0] * (grid[1] // 2) * (grid[2] // 2)
    rep = {"config": {"layers": layers, "controlnet_layers": cn_layers, "latent": [1, 16, *grid], "tokens": tokens, "steps": steps,
                      "taps_after_dit_block": [t + 1 for t in taps], "weights": ("random-init bf16 (bench.py seeds 100/200/300/400)" if PRETRAINED is None else f"TRAINED: {PRETRAINED}")
                                 + ("" if PEAKY == 1.0 else f", self-attention logits x {PEAKY:g}"),
                      "device": torch.cuda.get_device_name(0), "fp32_attention_q_chunk": q_chunk}}

    # ---- the product
    t0 = time.time()
    hip_lat, hip_first, hip_tap = hip_loop(pipe, inp, steps, taps)
    log(f"  hip: {steps} steps + tapped forward in {time.time() - t0:.1f} s")
    # ---- fp32 math (the yardstick)
    f32_tap = {}
    o32 = OracleRunner(pipe, cfg, torch.float32, q_chunk)
    f32_lat, f32_first = o32.loop(inp, steps, tap0=lambda i, x: f32_tap.__setitem__(i, x.clone()) if i in taps else None, log=log)
    # ---- the reference's own bf16 arithmetic
    b16_tap = {}
    o16 = OracleRunner(pipe, cfg, torch.bfloat16, q_chunk)
    b16_lat, b16_first = o16.loop(inp, steps, tap0=lambda i, x: b16_tap.__setitem__(i, x.clone()) if i in taps else None, log=log)

    def rows(name, tap, first, lats):
        r = {"after_block": {str(i + 1): rel_l2(tap[i].float(), f32_tap[i]) for i in taps if i in tap},
             "noise_pred_step0_cond": rel_l2(first.float(), f32_first),
             "latents_after_step": [rel_l2(a.float(), b) for a, b in zip(lats, f32_lat)]}
        log(f"  {name} vs fp32: blocks {r['after_block']}  noise_pred {r['noise_pred_step0_cond']:.3e}  "
            f"latents/step {[f'{e:.3e}' for e in r['latents_after_step']]}")
        return r

    rep["hip_bf16_vs_fp32"] = rows("hip-bf16", hip_tap, hip_first, hip_lat)
    rep["ref_bf16_vs_fp32"] = rows("ref-bf16", b16_tap, b16_first, b16_lat)
    rep["hip_bf16_vs_ref_bf16"] = {"noise_pred_step0_cond": rel_l2(hip_first.float(), b16_first.float()),
                                   "latents_after_step": [rel_l2(a.float(), b.float()) for a, b in zip(hip_lat, b16_lat)]}
    u8 = {}
    if decode:
        u8 = {"fp32": decode_u8(pipe, f32_lat[-1]), "hip": decode_u8(pipe, hip_lat[-1]), "ref": decode_u8(pipe, b16_lat[-1])}
        rep["psnr_db_decoded_uint8_frames"] = {"hip_bf16_vs_fp32": psnr_u8(u8["hip"], u8["fp32"]), "ref_bf16_vs_fp32": psnr_u8(u8["ref"], u8["fp32"]),
                                               "hip_bf16_vs_ref_bf16": psnr_u8(u8["hip"], u8["ref"]),
                                               "note": "all three final latents through the SAME tiled HIP VAE decode (random-init VAE) -> uint8 as UTIL:76-91"}
        log(f"  PSNR of decoded frames: {rep['psnr_db_decoded_uint8_frames']}")
    _dump(rep, out_path)            # the bf16 part is on disk before the fp8 part starts (a long run cut off late keeps it)
    if fp8:
        del b16_tap, o16
        for m in (pipe.dit, pipe.dit2, pipe.controlnet, pipe.controlnet2):
            if m is not None:
                enable_fp8(m)
        h8_lat, h8_first, h8_tap = hip_loop(pipe, inp, steps, taps)
        for m in (pipe.dit, pipe.dit2, pipe.controlnet, pipe.controlnet2):
            if m is not None:
                enable_fp8(m, False)
        c8_tap = {}
        o8 = OracleRunner(pipe, cfg, torch.bfloat16, q_chunk, fp8_chain=True)
        c8_lat, c8_first = o8.loop(inp, steps, tap0=lambda i, x: c8_tap.__setitem__(i, x.clone()) if i in taps else None, log=log)
        rep["hip_fp8_vs_fp32"] = rows("hip-fp8", h8_tap, h8_first, h8_lat)
        rep["scaled_mm_chain_vs_fp32"] = rows("scaled_mm-chain", c8_tap, c8_first, c8_lat)
        rep["hip_fp8_vs_scaled_mm_chain"] = {"noise_pred_step0_cond": rel_l2(h8_first.float(), c8_first.float()),
                                             "latents_after_step": [rel_l2(a.float(), b.float()) for a, b in zip(h8_lat, c8_lat)]}
        if decode:
            a, b = decode_u8(pipe, h8_lat[-1]), decode_u8(pipe, c8_lat[-1])
            rep["psnr_db_decoded_uint8_frames"].update({"hip_fp8_vs_fp32": psnr_u8(a, u8["fp32"]), "scaled_mm_chain_vs_fp32": psnr_u8(b, u8["fp32"]),
                                                        "hip_fp8_vs_scaled_mm_chain": psnr_u8(a, b)})
            log(f"  PSNR (fp8): {rep['psnr_db_decoded_uint8_frames']}")
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--cn-layers", type=int, default=10)
    ap.add_argument("--grid", type=int, nargs=3, default=[21, 60, 104], help="latent f, H/8, W/8 (default 832x480x81f)")
    ap.add_argument("--steps", type=int, default=4, help="0: one cond forward only (run_forward, what the gated full-size test runs)")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--q-chunk", type=int, default=2048)
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--peaky", type=float, default=1.0, help="multiply every self-attention's norm_q weight by this factor (logits x f): parity on "
                    "peaky softmax rows, the regime trained attention lives in")
    ap.add_argument("--models-dir", default="./models/Wan-AI", help="where the reference keeps its checkpoints (INF:81-106): when both experts' "
                    "shards and Wan2.1_VAE.pth are there the run uses TRAINED weights through from_pretrained, else random-init ones")
    ap.add_argument("--model-ckpt-path", default=None, help="ControlNet checkpoint step-N.safetensors (INF:51, 108); only with trained weights")
    a = ap.parse_args()
    global PRETRAINED, PEAKY
    PEAKY = a.peaky
    if pretrained_paths(a.models_dir) is not None:
        PRETRAINED = {"models_dir": a.models_dir, "ckpt": a.model_ckpt_path}
        print(f"fullsize_parity: TRAINED weights from {a.models_dir}" + (f" + ControlNet {a.model_ckpt_path}" if a.model_ckpt_path else
              " (ControlNet = copies of DiT blocks 0..N-1, zero zero-convs: GF:559-568)"), file=sys.stderr)
    else:
        print(f"fullsize_parity: no checkpoints under {a.models_dir}: random-init weights (trained-weight parity stays unpinned)", file=sys.stderr)
    t0 = time.time()
    if a.steps == 0:
        rep = run_forward(a.layers, a.cn_layers, tuple(a.grid), a.fp8, q_chunk=a.q_chunk, out_path=a.out)
    else:
        rep = run(a.layers, a.cn_layers, tuple(a.grid), a.steps, a.fp8, q_chunk=a.q_chunk, decode=not a.no_decode, out_path=a.out)
    rep["wall_s"] = time.time() - t0
    print(json.dumps(rep))
    _dump(rep, a.out)


if __name__ == "__main__":
    main()
