"""Pins the CPU oracle (oracle/wan_oracle.py) against the golden vectors produced by running the
reference's own modules (tests/golden/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN, rel_l2
from oracle import wan_oracle as wo

BF = torch.bfloat16
torch.set_grad_enabled(False)


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _bf(a):
    return gi.from_u16(a)


def test_scheduler_tables_bit_exact():
    g = _load("g1_scheduler.npz")
    for n in (50, 4, 3):
        s, t = wo.flow_match_sigmas(n, shift=5.0)
        assert np.array_equal(s.numpy(), g[f"sigmas_{n}"])
        assert np.array_equal(t.numpy(), g[f"timesteps_{n}"])
    s, t = wo.flow_match_sigmas(50, shift=5.0)
    assert int((t >= 875).sum()) == int(g["n_high_noise_50"]) == 21
    sample, mo = torch.from_numpy(g["sample"]), torch.from_numpy(g["model_output"])
    for i in (0, 10, 49):
        assert np.array_equal(wo.euler_step(mo, i, sample, s).numpy(), g[f"step_f32_{i}"])
        got = wo.euler_step(mo.to(BF), i, sample.to(BF), s)
        assert got.dtype == BF and torch.equal(got, _bf(g[f"step_bf16_{i}"]))


def _tiny_block():
    cfg = gi.TINY
    sd = gi.block_sd(torch.Generator().manual_seed(11), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 72, gi.TINY_CTX_LEN, seed=12)
    return cfg, sd, x, ctx, t_mod


def test_inputs_regenerate_identically():
    g = _load("g2_ops.npz")
    cfg, sd, x, ctx, t_mod = _tiny_block()
    assert gi.same_checksum(gi.checksum(sd), g["ck_weights"])
    assert gi.same_checksum(gi.checksum([x, ctx, t_mod]), g["ck_inputs"])


def test_ops_match_reference_bit_exact_on_cpu():
    g = _load("g2_ops.npz")
    cfg, sd, x, ctx, t_mod = _tiny_block()
    nh, eps = cfg["num_heads"], cfg["eps"]
    ts = torch.tensor([995.9], dtype=BF)
    assert torch.equal(wo.sinusoidal_embedding_1d(256, ts), _bf(g["sinus_bf16"]))
    assert np.array_equal(wo.sinusoidal_embedding_1d(256, ts.float()).numpy(), g["sinus_f32"])
    freqs = wo.rope_freqs_3d(cfg["dim"] // nh, 3, 4, 6)
    assert np.array_equal(freqs.real.numpy(), g["freqs_re"]) and np.array_equal(freqs.imag.numpy(), g["freqs_im"])
    assert torch.equal(wo.rope_apply(x, freqs, nh), _bf(g["rope_bf16"]))
    for mode, dt in (("bf16", BF), ("f32", torch.float32)):
        s = {k: v.to(dt) for k, v in sd.items()}
        xi, ci, ti = x.to(dt), ctx.to(dt), t_mod.to(dt)

        def same(got, key):
            ref = _bf(g[key]) if dt == BF else torch.from_numpy(g[key])
            if dt == BF:
                assert torch.equal(got, ref), key
            else:  # fp32: same math, possibly different op grouping
                assert rel_l2(got, ref) < 2e-6, key

        same(wo.rms_norm(xi, s["self_attn.norm_q.weight"], eps), f"rmsnorm_{mode}")
        mod = (s["modulation"].to(dt) + ti).chunk(6, dim=1)
        if dt == BF:
            # plain nn.LayerNorm (bf16 in/out) vs fp32-LN-then-cast: equal up to 1 bf16 ulp on rare ties
            got = wo.modulate(wo.layer_norm(xi, eps=eps), mod[0], mod[1])
            assert rel_l2(got, _bf(g[f"ln_mod_{mode}"])) < 2e-3
            assert torch.equal(wo.layer_norm(xi, eps=eps), _bf(g["ln_autocast_bf16"]))
            assert rel_l2(wo.layer_norm(xi, s["norm3.weight"], s["norm3.bias"], eps), _bf(g["ln_affine_bf16"])) < 2e-3
        else:
            same(wo.modulate(wo.layer_norm(xi, eps=eps), mod[0], mod[1]), f"ln_mod_{mode}")
            same(wo.layer_norm(xi, s["norm3.weight"], s["norm3.bias"], eps), f"ln_affine_{mode}")
        tol = 3e-3 if dt == BF else 2e-6

        def close(got, key):
            ref = _bf(g[key]) if dt == BF else torch.from_numpy(g[key])
            assert rel_l2(got, ref) < tol, (key, rel_l2(got, ref))

        close(wo.self_attention(xi, freqs, s, "self_attn.", nh, eps), f"self_attn_{mode}")
        close(wo.cross_attention(xi, ci, s, "cross_attn.", nh, eps), f"cross_attn_{mode}")
        close(F_ffn(xi, s), f"ffn_{mode}")
        close(wo.dit_block(xi, ci, ti, freqs, s, "", nh, eps), f"block_{mode}")


def F_ffn(x, s):
    import torch.nn.functional as F
    return F.linear(F.gelu(F.linear(x, s["ffn.0.weight"], s["ffn.0.bias"]), approximate="tanh"),
                    s["ffn.2.weight"], s["ffn.2.bias"])


def _tiny_models(zero):
    return gi.dit_sd(gi.TINY, seed=41), gi.controlnet_sd(gi.TINY, gi.TINY_CONTROLNET_LAYERS, seed=42,
                                                         zero_convs_zero=zero)


@pytest.mark.parametrize("mode", ["bf16", "f32"])
def test_model_fn_and_loop_match_reference(mode):
    g = _load("g5_model_fn.npz")
    dt = BF if mode == "bf16" else torch.float32
    inp = gi.tiny_inputs()
    assert gi.same_checksum(gi.checksum(inp), g["ck_inputs"])
    tol = 5e-3 if dt == BF else 5e-6
    cfg = dict(gi.TINY)
    ts = torch.tensor([995.9], dtype=BF).to(dt)

    def ref(key):
        return _bf(g[key]) if dt == BF else torch.from_numpy(g[key])

    def c(sd):
        return {k: v.to(dt) for k, v in sd.items()}

    for zero, tag in ((False, "rand"), (True, "zero")):
        dsd, csd = _tiny_models(zero)
        if not zero and dt == BF:
            assert gi.same_checksum(gi.checksum(dsd), g["ck_dit"]) and gi.same_checksum(gi.checksum(csd), g["ck_controlnet"])
        out = wo.model_fn(c(dsd), cfg, inp["latents"].to(dt), ts, inp["ctx_posi"].to(dt), inp["y"].to(dt), c(csd),
                          inp["control"].to(dt), gi.TINY_CONTROLNET_LAYERS)
        assert rel_l2(out, ref(f"model_fn_cn_{tag}_{mode}")) < tol
        if zero:
            # property: a zero zero-conv makes the ControlNet a bitwise no-op
            out2 = wo.model_fn(c(dsd), cfg, inp["latents"].to(dt), ts, inp["ctx_posi"].to(dt), inp["y"].to(dt))
            assert torch.equal(out, out2)
            assert rel_l2(out2, ref(f"model_fn_nocn_{mode}")) < tol
    dsd1, csd1 = _tiny_models(False)
    dsd2, csd2 = gi.dit_sd(gi.TINY, seed=43), gi.controlnet_sd(gi.TINY, 1, seed=42, zero_convs_zero=True)
    lat = wo.denoise_loop([(c(dsd1), cfg, c(csd1), 1), (c(dsd2), cfg, c(csd2), 1)], inp["latents"].to(dt),
                          inp["ctx_posi"].to(dt), inp["ctx_nega"].to(dt), inp["y"].to(dt), inp["control"].to(dt), 3,
                          dtype=dt)
    assert [bool(v) for v in g["loop3_switched"]] == [False, False, True]  # 1000, 909.1, 714.3 vs 875
    assert rel_l2(lat, ref(f"loop3_{mode}")) < (2e-2 if dt == BF else 2e-5)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_oracle_pipeline_call_vs_reference_call(mode):
    """oracle/pipeline_oracle.py::pipeline_call against g13 = the outputs of the reference's OWN `WanVideoPipeline.__call__`
    (GF:598-737, executed by make_goldens.py::g13_pipeline_call): unit order, noise, both embedders, the 3-step loop with the
    expert switch and CFG, the tiled decode, the uint8 frames.  fp32: the same arithmetic in another op order (1e-4, frames within
    one level); bf16: the bar of the other bf16 loop golden (g5 loop3: 2e-2 ... the loop amplifies, so 2 x that here with the VAE
    encodes in front) — and the post-loop chain alone, fed the reference's latents, bit for bit."""
    from oracle import pipeline_oracle as plo
    g = np.load(os.path.join(GOLDEN, "g13_pipeline_call.npz"))
    g6 = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    dt = torch.float32 if mode == "f32" else BF
    vsd = {k: v.to(dt) for k, v in gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61).items()}
    image, control = gi.preloop_inputs()
    inp = gi.tiny_inputs()
    assert gi.same_checksum(gi.checksum([torch.from_numpy(np.array(image)).float(), control, inp["ctx_posi"], inp["ctx_nega"]]), g["ck_inputs"])
    c = lambda sd: {k: v.to(dt) for k, v in sd.items()}
    cfg = dict(gi.TINY)
    nl = gi.TINY_CONTROLNET_LAYERS
    experts = [(c(gi.dit_sd(cfg, seed=41)), cfg, c(gi.controlnet_sd(cfg, nl, seed=42)), nl),
               (c(gi.dit_sd(cfg, seed=43)), cfg, c(gi.controlnet_sd(cfg, nl, seed=42, zero_convs_zero=True)), nl)]
    asked = []

    def encode_prompt(p):
        asked.append(p)
        return inp["ctx_posi"] if p == gi.PIPELINE_PROMPTS[0] else inp["ctx_nega"]
    kw = {k: v for k, v in gi.PIPELINE_CALL_KWARGS.items() if k != "controlnet"}
    lat, video, u8 = plo.pipeline_call(experts, vsd, encode_prompt, gi.PIPELINE_PROMPTS[0], gi.PIPELINE_PROMPTS[1], image, control,
                                       dtype=dt, **kw)
    assert asked == list(gi.PIPELINE_PROMPTS)
    ref_lat = torch.from_numpy(g["latents_f32"]) if mode == "f32" else gi.from_u16(g["latents_bf16"])
    ref_u8 = g[f"frames_u8_{mode}"]
    e = rel_l2(lat.float(), ref_lat.float())
    d = np.abs(u8.numpy().astype(np.int32) - ref_u8.astype(np.int32))
    if mode == "f32":
        assert e < 1e-4, e
        assert int(d.max()) <= 1 and float((d == 0).mean()) > 0.995, (int(d.max()), float((d == 0).mean()))
    else:
        assert e < 4e-2, e
        # the post-loop chain alone on the reference's own final latents: same torch ops, same bytes
        z = gi.from_u16(g["latents_bf16"])
        v2 = vo_tiled(z, vsd, kw["tile_size"], kw["tile_stride"])
        assert torch.equal(v2, gi.from_u16(g["video_bf16"]))
        assert np.array_equal(plo.frames_uint8(v2).numpy(), g["frames_u8_bf16"])
    assert u8.shape == (9, 64, 96, 3) and tuple(video.shape) == (1, 3, 9, 64, 96)


def vo_tiled(z, sd, tile_size, tile_stride):
    from oracle import vae_oracle as vo
    return vo.tiled_decode(z, sd, tile_size, tile_stride)
