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


def test_b3_module_level_functions_vs_reference_goldens():
    """the reference's module-level names with its signatures — `rope_apply(x, freqs, num_heads)` (DIT:92),
    `modulate(x, shift, scale)` (DIT:64), `flash_attention(q, k, v, num_heads, compatibility_mode=False)` (DIT:28) — exported by
    goal_force_amd.dit and run on the HIP kernels, against what the reference's own functions produced (g2)."""
    from goal_force_amd import ops
    from goal_force_amd.dit import flash_attention, modulate, rope_apply
    g = _load("g2_ops.npz")
    cfg = gi.TINY
    nh = cfg["num_heads"]
    sd = gi.block_sd(torch.Generator().manual_seed(11), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 72, gi.TINY_CTX_LEN, seed=12)
    xg = x.cuda()
    # rope_apply with the reference's complex [S, 1, d/2] table: one bf16 rounding of an exact rotation -> <= 1 ulp from the golden
    freqs = torch.complex(torch.from_numpy(g["freqs_re"]), torch.from_numpy(g["freqs_im"])).reshape(72, 1, 64)
    got = rope_apply(xg, freqs, nh).cpu()
    ref = _bf(g["rope_bf16"])
    assert got.shape == ref.shape
    d = (got.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3 and rel_l2(got.float(), torch.from_numpy(g["rope_f32"])) < 3e-3
    # modulate(norm1(x), shift, scale) == the fused LayerNorm+modulate kernel's output, and the reference's bf16 result
    mod = (sd["modulation"] + t_mod).chunk(6, dim=1)                       # bf16 add, as DIT:218
    ln = ops.layernorm_modulate(xg[0], eps=cfg["eps"]).unsqueeze(0)
    got = modulate(ln, mod[0].cuda(), mod[1].cuda())
    fused = ops.layernorm_modulate(xg[0], scale1p=(1 + mod[1]).reshape(-1).cuda(), shift=mod[0].reshape(-1).contiguous().cuda(), eps=cfg["eps"])
    assert torch.equal(got[0], fused)
    ref = _bf(g["ln_mod_bf16"])
    d = (got.cpu().view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    assert float((d > 1).float().mean()) < 2e-3 and rel_l2(got.cpu().float(), torch.from_numpy(g["ln_mod_f32"])) < 4e-3
    # flash_attention on [B, S, n d] with B = 2 (two different batches) vs the fp64 attention of the oracle
    gq = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn((2, 300, nh * 128), generator=gq).to(BF) for _ in range(3))
    got = flash_attention(q.cuda(), k[:, :200].cuda(), v[:, :200].cuda(), num_heads=nh)
    got_c = flash_attention(q.cuda(), k[:, :200].cuda(), v[:, :200].cuda(), nh, compatibility_mode=True)
    assert tuple(got.shape) == (2, 300, nh * 128) and torch.equal(got, got_c)
    ref = wo.attention_fp64(q, k[:, :200], v[:, :200], nh)
    assert rel_l2(got.cpu().float(), ref.float()) < 4e-3


def test_b3_flash_attention_on_a_finished_q_keeps_sdpa_precision_at_long_peaky_keys():
    """DIT:28-61 hands SDPA a q that is already bf16: at >= 2048 keys the drop-in runs on the kernel that scales the fp32 scores
    (finished_q), within 1.4 x torch's SDPA from fp64 (measured 1.12-1.20 x) at logit std 3 and 8 — where kernel 3 on the same finished q, which rounds
    Q x scale x log2(e) to bf16 again, is 2-3 x further (the package's blocks feed kernel 3 a q that was scaled BEFORE its only rounding)."""
    import math
    import torch.nn.functional as F
    from goal_force_amd import ops
    from goal_force_amd.dit import flash_attention
    S, H, HD = ops.VT_MIN_KV + 256, 4, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    for qs in (3.0, 8.0):
        q = (torch.randn((1, S, H * HD), generator=g, device="cuda") * qs).to(BF)
        k, v = (torch.randn((1, S, H * HD), generator=g, device="cuda").to(BF) for _ in range(2))
        heads = lambda t: t[0].view(S, H, HD).transpose(0, 1)
        ref = (torch.softmax(heads(q).double() @ heads(k).double().transpose(1, 2) / math.sqrt(HD), -1) @ heads(v).double()).transpose(0, 1).reshape(S, H * HD)
        rel = lambda a: float((a.double() - ref).norm() / ref.norm())
        sdpa = rel(F.scaled_dot_product_attention(heads(q)[None], heads(k)[None], heads(v)[None])[0].transpose(0, 1).reshape(S, H * HD))
        b3 = rel(flash_attention(q, k, v, H)[0])
        k3 = rel(ops.flash_attn(q[0], k[0], v[0], H))
        print(f"finished q, {S} keys, logit std {qs:g}: B3 flash_attention {b3:.2e}  kernel 3 {k3:.2e}  torch SDPA {sdpa:.2e}")
        assert b3 <= 1.4 * sdpa, (qs, b3, sdpa)
        assert k3 > 1.6 * b3, (qs, k3, b3)


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


def test_cfg_pair_shares_block0_self_attention_bit_identically():
    """model_fn `cfg_shared`: the second forward of a CFG step takes the context-independent half of block 0 (DiT and
    ControlNet) from the first.  Same bits as two independent forwards, for both branch orders; the dict is drained."""
    from goal_force_amd.model_fn import model_fn_wan_video
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    ts = torch.tensor([995.9], dtype=BF).cuda()
    dit, cn = _tiny_pipeline(False)
    kw = dict(latents=inp["latents"], timestep=ts, y=inp["y"], controlnet=cn, control_signal_video_latents=inp["control"],
              elide_zero_controlnet=False)
    want_p = model_fn_wan_video(dit, context=inp["ctx_posi"], **kw)
    want_n = model_fn_wan_video(dit, context=inp["ctx_nega"], **kw)
    assert not torch.equal(want_p, want_n)
    for first, second, w1, w2 in (("ctx_posi", "ctx_nega", want_p, want_n), ("ctx_nega", "ctx_posi", want_n, want_p)):
        pair = {}
        got1 = model_fn_wan_video(dit, context=inp[first], cfg_shared=pair, **kw)
        assert set(pair) == {"cn0", "dit0"} and all("x" in m for m in pair.values())
        got2 = model_fn_wan_video(dit, context=inp[second], cfg_shared=pair, **kw)
        assert torch.equal(got1, w1) and torch.equal(got2, w2)
        assert all(len(m) == 0 for m in pair.values())
    # without a ControlNet only the DiT's block 0 is shared
    pair = {}
    a = model_fn_wan_video(dit, latents=inp["latents"], timestep=ts, y=inp["y"], context=inp["ctx_posi"], cfg_shared=pair)
    b = model_fn_wan_video(dit, latents=inp["latents"], timestep=ts, y=inp["y"], context=inp["ctx_nega"], cfg_shared=pair)
    assert torch.equal(a, model_fn_wan_video(dit, latents=inp["latents"], timestep=ts, y=inp["y"], context=inp["ctx_posi"]))
    assert torch.equal(b, model_fn_wan_video(dit, latents=inp["latents"], timestep=ts, y=inp["y"], context=inp["ctx_nega"]))


def test_denoise_loop_same_bits_with_and_without_shared_prefix():
    from goal_force_amd.pipeline import WanVideoPipeline
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    dit1, cn1 = _tiny_pipeline(False)
    dit2, cn2 = _tiny_pipeline(True, dit_seed=43)
    pipe = WanVideoPipeline.from_modules(dit1, dit2, cn1, cn2)
    run = lambda: pipe.denoise(inp["latents"], inp["ctx_posi"], inp["ctx_nega"], inp["y"], inp["control"],
                               num_inference_steps=3, cfg_scale=5.0, controlnet=True)
    assert pipe.share_cfg_prefix
    a = run()
    pipe.share_cfg_prefix = False
    assert torch.equal(a, run())


def test_denoise_loop_same_bits_with_the_cfg_pair_on_two_streams():
    """pipeline.cfg_streams (GF_CFG_STREAMS=1): the uncond forward on a second HIP stream, ordered after the cond forward's block-0
    halves by events — an A/B switch (measured slower at full size), but it must produce the same bits."""
    from goal_force_amd.pipeline import WanVideoPipeline
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    dit1, cn1 = _tiny_pipeline(False)
    dit2, cn2 = _tiny_pipeline(True, dit_seed=43)
    pipe = WanVideoPipeline.from_modules(dit1, dit2, cn1, cn2)
    run = lambda: pipe.denoise(inp["latents"], inp["ctx_posi"], inp["ctx_nega"], inp["y"], inp["control"],
                               num_inference_steps=3, cfg_scale=5.0, controlnet=True)
    assert not pipe.cfg_streams
    a = run()
    pipe.cfg_streams = True
    for _ in range(3):
        assert torch.equal(a, run())
    assert pipe._cfg_side_stream is not None
    torch.cuda.synchronize()


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


# (Round 6: the S = 32760 single-block sampled-rows tests that stood here — bf16 against an fp32 chain, fp8 against a live
# torch._scaled_mm chain — are subsumed by tests/test_fulldepth_gpu.py::test_full_size_forward_bf16_and_fp8_within_reference_drift,
# which runs the same comparisons through all 40 + 10 blocks at S = 32760, and by test_a14b_block_config1_vs_golden above.)


def test_fp8_model_fn_and_loop_vs_fp8_oracle():
    """BASELINE config 5 above block level: tiny model_fn with ControlNet and the 3-step CFG loop with the expert
    switch, every Linear inside the DiT / ControlNet blocks on the fp8_linear contract (VRAM:115-151) — HIP against the
    oracle's model_fn with the restated fp8_linear swapped in for exactly those Linears."""
    import torch.nn.functional as F
    from oracle import fp8_oracle as fo
    from goal_force_amd.dit import enable_fp8
    from goal_force_amd.model_fn import model_fn_wan_video
    from goal_force_amd.pipeline import WanVideoPipeline
    cfg = gi.TINY
    inp = gi.tiny_inputs()
    dev = {k: v.cuda() for k, v in inp.items()}
    sds = [(gi.dit_sd(cfg, seed=41), gi.controlnet_sd(cfg, 1, seed=42)),
           (gi.dit_sd(cfg, seed=43), gi.controlnet_sd(cfg, 1, seed=42, zero_convs_zero=True))]
    block_w = {id(t) for dsd, csd in sds for sd_ in (dsd, csd) for k, t in sd_.items()
               if ("blocks." in k) and k.endswith(".weight") and t.dim() == 2}

    def lin(x, w, b=None):
        return fo.fp8_linear(x, w, b) if id(w) in block_w else F.linear(x, w, b)

    ts = torch.tensor([995.9], dtype=BF)
    old = wo.LINEAR
    try:
        exact = wo.model_fn(sds[0][0], cfg, inp["latents"], ts, inp["ctx_posi"], inp["y"], sds[0][1], inp["control"], 1)
        wo.LINEAR = lin
        ref = wo.model_fn(sds[0][0], cfg, inp["latents"], ts, inp["ctx_posi"], inp["y"], sds[0][1], inp["control"], 1)
        ref_loop = wo.denoise_loop([(sds[0][0], cfg, sds[0][1], 1), (sds[1][0], cfg, sds[1][1], 1)], inp["latents"],
                                   inp["ctx_posi"], inp["ctx_nega"], inp["y"], inp["control"], 3)
    finally:
        wo.LINEAR = old
    dit1, cn1 = _tiny_pipeline(False)
    dit2, cn2 = _tiny_pipeline(True, dit_seed=43)
    for m in (dit1, cn1, dit2, cn2):
        enable_fp8(m)
    got = model_fn_wan_video(dit1, latents=dev["latents"], timestep=ts.cuda(), context=dev["ctx_posi"], y=dev["y"],
                             controlnet=cn1, control_signal_video_latents=dev["control"]).cpu()
    e, e_q = rel_l2(got.float(), ref.float()), rel_l2(ref.float(), exact.float())
    assert e_q > 5e-3, "the fp8 contract must actually be in force in the oracle run"
    assert e < 0.5 * e_q, f"fp8 model_fn: HIP vs fp8 oracle {e:.3e} (fp8 contract vs bf16 graph {e_q:.3e})"
    pipe = WanVideoPipeline.from_modules(dit1, dit2, cn1, cn2)
    lat = pipe.denoise(dev["latents"], dev["ctx_posi"], dev["ctx_nega"], dev["y"], dev["control"],
                       num_inference_steps=3, cfg_scale=5.0, controlnet=True).cpu()
    g = _load("g5_model_fn.npz")
    f32 = torch.from_numpy(g["loop3_f32"])
    e_loop, e_ref = rel_l2(lat.float(), ref_loop.float()), rel_l2(ref_loop.float(), f32)
    # CFG x5 over 3 steps amplifies every rounding difference (the bf16 reference itself is 0.153 from fp32 math here):
    # the two fp8 runs must be closer to each other than the fp8 oracle run is to exact math
    assert e_loop < e_ref, f"fp8 3-step loop: HIP vs fp8 oracle {e_loop:.3e} (fp8 oracle vs fp32 math {e_ref:.3e})"


def test_time_embedding_sinusoid_on_device_matches_reference_golden():
    """DIT:68-72 computed on the timestep's device (no host round trip): bit-identical to the reference's CPU result for the
    fixture's timesteps and for every timestep of the 50-step schedule (bf16-rounded as the pipeline passes it, GF:707)."""
    from goal_force_amd.dit import sinusoidal_embedding_1d
    from goal_force_amd.scheduler import FlowMatchScheduler
    g = _load("g2_ops.npz")
    ts = torch.tensor([995.9], dtype=BF)                       # the fixture's timestep (make_goldens.py::g2_ops)
    got = sinusoidal_embedding_1d(256, ts.cuda())
    assert got.is_cuda and torch.equal(got.cpu(), _bf(g["sinus_bf16"]))
    sch = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(50, shift=5.0)
    t50 = sch.timesteps.to(BF)
    assert torch.equal(sinusoidal_embedding_1d(256, t50.cuda()).cpu(), wo.sinusoidal_embedding_1d(256, t50))
