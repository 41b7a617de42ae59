"""GPU parity of the composite hot path (DiT block, model_fn with ControlNet, CFG loop) against
 (a) the golden vectors produced by the reference's own modules and (b) the CPU oracle on the same
seeded inputs.  Tolerances from SURVEY.md §8(d): per-block rel-L2 <= 5e-3 vs the fp32-math result and
<= 2x the reference-bf16's own error; 3-step tiny pipeline latents rel-L2 <= 2e-2.
"""
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN, rel_l2
from oracle import wan_oracle as wo

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
torch.set_grad_enabled(False)


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _bf(a):
    return gi.from_u16(a)


def _block(cfg, sd):
    from goal_force_amd.dit import DiTBlock
    blk = DiTBlock(False, cfg["dim"], cfg["num_heads"], cfg["ffn_dim"], cfg["eps"])
    blk.load_state_dict(sd, strict=True)
    return blk.to(BF).cuda()


def _rope(cfg, grid):
    from goal_force_amd.dit import RopeTable, precompute_freqs_cis_3d
    return RopeTable.from_grid(precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"]), *grid, "cuda")


def test_tiny_block_and_submodules_vs_goldens():
    g = _load("g2_ops.npz")
    cfg = gi.TINY
    sd = gi.block_sd(torch.Generator().manual_seed(11), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 72, gi.TINY_CTX_LEN, seed=12)
    assert gi.same_checksum(gi.checksum(sd), g["ck_weights"]) and gi.same_checksum(gi.checksum([x, ctx, t_mod]), g["ck_inputs"])
    blk = _block(cfg, sd)
    rope = _rope(cfg, (3, 4, 6))
    xg, cg, tg = x.cuda(), ctx.cuda(), t_mod.cuda()
    res = {
        "self_attn": blk.self_attn(xg, rope),
        "cross_attn": blk.cross_attn(xg, cg),
        "block": blk(xg, cg, tg, rope),
    }
    from goal_force_amd import ops
    f1 = ops.gemm(xg[0], blk.ffn[0].weight, blk.ffn[0].bias, epilogue=ops.EPI_BIAS_GELU_TANH)
    res["ffn"] = ops.gemm(f1, blk.ffn[2].weight, blk.ffn[2].bias).unsqueeze(0)
    for k, got in res.items():
        f32 = torch.from_numpy(g[f"{k}_f32"])
        ref_bf = _bf(g[f"{k}_bf16"])
        e = rel_l2(got.cpu().float(), f32)
        e_ref = rel_l2(ref_bf.float(), f32)
        # bar: <= 5e-3 for the block, and never worse than 1.25x the reference-bf16's own error vs fp32 math
        # (at this tiny width the reference's self-attention alone already sits at 5.3e-3)
        lim = 5e-3 if k == "block" else 1e-2
        assert e < lim and e < 1.25 * e_ref + 1e-4, f"{k}: rel_l2 vs fp32 {e:.3e} (reference bf16 itself {e_ref:.3e})"
    # the block must not modify its input
    assert torch.equal(xg.cpu(), x)


def test_block_accepts_reference_complex_freqs():
    cfg = gi.TINY
    sd = gi.block_sd(torch.Generator().manual_seed(11), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 72, gi.TINY_CTX_LEN, seed=12)
    blk = _block(cfg, sd)
    freqs = wo.rope_freqs_3d(128, 3, 4, 6).reshape(72, 1, 64)  # what the reference passes (GF:1474-1478)
    a = blk(x.cuda(), ctx.cuda(), t_mod.cuda(), freqs)
    b = blk(x.cuda(), ctx.cuda(), t_mod.cuda(), _rope(cfg, (3, 4, 6)))
    assert torch.equal(a, b)


def _check_sparse(got, g, tag):
    rows = torch.from_numpy(g["rows"])
    gc = got.cpu().float()[0]
    ref_rows = torch.from_numpy(g["bf16_rows"])
    e_rows = rel_l2(gc[rows], ref_rows)
    e_norm = rel_l2(gc.norm(dim=-1), torch.from_numpy(g["bf16_norms"]))
    return e_rows, e_norm


def test_mid_block_vs_golden():
    g = _load("g3_block_mid.npz")
    cfg = gi.MID
    grid = (5, 12, 16)
    sd = gi.block_sd(torch.Generator().manual_seed(21), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 960, 512, seed=22)
    assert gi.same_checksum(gi.checksum(sd), g["ck_weights"]) and gi.same_checksum(gi.checksum([x, ctx, t_mod]), g["ck_inputs"])
    got = _block(cfg, sd)(x.cuda(), ctx.cuda(), t_mod.cuda(), _rope(cfg, grid))
    rows = torch.from_numpy(g["rows"])
    e32 = rel_l2(got.cpu().float()[0][rows], torch.from_numpy(g["f32_rows"]))
    e_ref = float(g["ref_bf16_vs_f32_rel_l2"])
    e_rows, e_norm = _check_sparse(got, g, "mid")
    assert e32 < 5e-3 and e32 < 2 * e_ref, f"vs fp32 {e32:.3e} (reference bf16 {e_ref:.3e})"
    assert e_rows < 8e-3 and e_norm < 2e-3, f"vs reference bf16 rows {e_rows:.3e} norms {e_norm:.3e}"


def test_a14b_block_config1_vs_golden():
    """BASELINE config 1: one A14B-dims DiT block on the 9x30x52 grid (S=14040)."""
    g = _load("g4_block_a14b.npz")
    cfg = gi.A14B
    grid = (9, 30, 52)
    sd = gi.block_sd(torch.Generator().manual_seed(31), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 14040, 512, seed=32)
    assert gi.same_checksum(gi.checksum(sd), g["ck_weights"]) and gi.same_checksum(gi.checksum([x, ctx, t_mod]), g["ck_inputs"])
    got = _block(cfg, sd)(x.cuda(), ctx.cuda(), t_mod.cuda(), _rope(cfg, grid))
    e_rows, e_norm = _check_sparse(got, g, "a14b")
    # both sides carry ~3e-3 of bf16 noise vs exact math, so their mutual distance is ~sqrt(2) of that
    assert e_rows < 8e-3 and e_norm < 2e-3, f"vs reference bf16 rows {e_rows:.3e} norms {e_norm:.3e}"


def _tiny_pipeline(zero_cn, dit_seed=41):
    from goal_force_amd.controlnet import ControlNet
    from goal_force_amd.dit import WanModel
    cfg = gi.TINY
    dit = WanModel(has_image_input=False, require_clip_embedding=False, **cfg)
    dit.load_state_dict(gi.dit_sd(cfg, seed=dit_seed), strict=True)
    cn = ControlNet(gi.TINY_CONTROLNET_LAYERS, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"])
    cn.load_state_dict(gi.controlnet_sd(cfg, gi.TINY_CONTROLNET_LAYERS, seed=42, zero_convs_zero=zero_cn), strict=True)
    return dit.to(BF).cuda(), cn.to(BF).cuda()


def test_model_fn_with_controlnet_vs_goldens():
    from goal_force_amd.model_fn import model_fn_wan_video
    g = _load("g5_model_fn.npz")
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    ts = torch.tensor([995.9], dtype=BF).cuda()
    for zero, tag in ((False, "rand"), (True, "zero")):
        dit, cn = _tiny_pipeline(zero)
        out = model_fn_wan_video(dit, latents=inp["latents"], timestep=ts, context=inp["ctx_posi"], y=inp["y"],
                                 controlnet=cn, control_signal_video_latents=inp["control"],
                                 elide_zero_controlnet=False)
        f32 = torch.from_numpy(g[f"model_fn_cn_{tag}_f32"])
        e = rel_l2(out.cpu().float(), f32)
        e_ref = rel_l2(_bf(g[f"model_fn_cn_{tag}_bf16"]).float(), f32)
        assert out.shape == (1, 16, 3, 8, 12)
        assert e < 1e-2 and e < 2 * e_ref + 1e-4, f"{tag}: vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
        if zero:
            # property: zero zero-convs => ControlNet is a bitwise no-op, so eliding it changes nothing
            out_elided = model_fn_wan_video(dit, latents=inp["latents"], timestep=ts, context=inp["ctx_posi"],
                                            y=inp["y"], controlnet=cn, control_signal_video_latents=inp["control"])
            out_none = model_fn_wan_video(dit, latents=inp["latents"], timestep=ts, context=inp["ctx_posi"], y=inp["y"])
            assert cn.all_zero() and torch.equal(out, out_elided) and torch.equal(out, out_none)


def test_three_step_cfg_loop_vs_golden():
    """GF:697-723 at tiny size: expert switch after step 2, CFG 5.0, Euler update; final latents."""
    from goal_force_amd.pipeline import WanVideoPipeline
    g = _load("g5_model_fn.npz")
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    dit1, cn1 = _tiny_pipeline(False)
    dit2, cn2 = _tiny_pipeline(True, dit_seed=43)
    pipe = WanVideoPipeline.from_modules(dit1, dit2, cn1, cn2)
    lat = pipe.denoise(inp["latents"], inp["ctx_posi"], inp["ctx_nega"], inp["y"], inp["control"],
                       num_inference_steps=3, cfg_scale=5.0, controlnet=True)
    f32 = torch.from_numpy(g["loop3_f32"])
    e = rel_l2(lat.cpu().float(), f32)
    e_ref = rel_l2(_bf(g["loop3_bf16"]).float(), f32)
    # CFG x5 over 3 steps of a random tiny model amplifies bf16 noise: the reference's own bf16 run is 0.153
    # away from its fp32 run, so the SURVEY's 2e-2 guess is unattainable by the reference itself.  Bar: not
    # measurably worse than the reference (<= 1.25x its own error) and within 2x of it from the bf16 golden.
    e_bf = rel_l2(lat.cpu().float(), _bf(g["loop3_bf16"]).float())
    assert e < 1.25 * e_ref + 1e-3 and e_bf < 2 * e_ref, f"vs fp32 {e:.3e}, vs ref-bf16 {e_bf:.3e} (reference bf16 vs fp32 {e_ref:.3e})"
    # context K/V caching and ControlNet2 elision must not change a single bit
    pipe.elide_zero_controlnet = False
    lat2 = pipe.denoise(inp["latents"], inp["ctx_posi"], inp["ctx_nega"], inp["y"], inp["control"],
                        num_inference_steps=3, cfg_scale=5.0, controlnet=True)
    assert torch.equal(lat, lat2)
    assert torch.equal(inp["latents"].cpu(), gi.tiny_inputs()["latents"])  # input untouched


def test_cfg_scale_one_is_single_forward():
    from goal_force_amd.pipeline import WanVideoPipeline
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    dit1, cn1 = _tiny_pipeline(False)
    pipe = WanVideoPipeline.from_modules(dit1, None, cn1, None)
    calls = []
    orig = pipe.model_fn

    def counting(**kw):
        calls.append(1)
        return orig(**kw)

    pipe.model_fn = counting
    pipe.denoise(inp["latents"], inp["ctx_posi"], None, inp["y"], inp["control"], num_inference_steps=2,
                 cfg_scale=1.0, controlnet=True)
    assert len(calls) == 2
