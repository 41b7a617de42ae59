"""B1 end to end against the REFERENCE's own `WanVideoPipeline.__call__` (GF:598-737).

tests/golden/g13_pipeline_call.npz was produced by executing the reference's `__call__` as it stands (its 17 units through its own
PipelineUnitRunner, its loop, its tiled decode, its `vae_output_to_video`) on tiny seeded experts + ControlNets, the real Wan VAE with
the g6 weights and a fixed-tensor prompter (tests/golden/make_goldens.py::g13_pipeline_call).  Here the product `pipe(...)` gets the
SAME keyword arguments (gen_inputs.PIPELINE_CALL_KWARGS: the reference's own names) and the same kind of prompter object."""
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def psnr_u8(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))


class FixedPrompter:
    """What the golden run used in place of WanPrompter: `encode_prompt` with the reference's signature (GF:819)."""

    def __init__(self, inp):
        self.inp, self.calls = inp, []

    def encode_prompt(self, prompt, positive=True, device="cuda"):
        self.calls.append((prompt, positive))
        return (self.inp["ctx_posi"] if prompt == gi.PIPELINE_PROMPTS[0] else self.inp["ctx_nega"]).to(device)


def _pipe():
    from goal_force_amd.controlnet import ControlNet
    from goal_force_amd.dit import WanModel
    from goal_force_amd.pipeline import WanVideoPipeline
    from goal_force_amd.vae import WanVideoVAE
    g = np.load(os.path.join(GOLDEN, "g13_pipeline_call.npz"))
    cfg = gi.TINY

    def expert(seed):
        m = WanModel(has_image_input=False, require_clip_embedding=False, **cfg)
        m.load_state_dict(gi.dit_sd(cfg, seed=seed), strict=True)
        return m.to(BF).cuda()

    def cnet(zero):
        cn = ControlNet(gi.TINY_CONTROLNET_LAYERS, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"])
        cn.load_state_dict(gi.controlnet_sd(cfg, gi.TINY_CONTROLNET_LAYERS, seed=42, zero_convs_zero=zero), strict=True)
        return cn.to(BF).cuda()

    assert gi.same_checksum(gi.checksum(gi.dit_sd(cfg, seed=41)), g["ck_dit"])
    assert gi.same_checksum(gi.checksum(gi.controlnet_sd(cfg, gi.TINY_CONTROLNET_LAYERS, seed=42)), g["ck_controlnet"])
    g6 = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    vsd = gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61)
    vae = WanVideoVAE()
    vae.load_state_dict({"model." + k: t for k, t in vsd.items()}, strict=True)
    pipe = WanVideoPipeline.from_modules(expert(41), expert(43), cnet(False), cnet(True), vae=vae.to(BF).cuda())
    image, control = gi.preloop_inputs()
    inp = gi.tiny_inputs()
    assert gi.same_checksum(gi.checksum([torch.from_numpy(np.array(image)).float(), control, inp["ctx_posi"], inp["ctx_nega"]]), g["ck_inputs"])
    assert str(g["kwargs_repr"]) == repr(sorted(gi.PIPELINE_CALL_KWARGS.items())), "the golden was made with these keyword arguments"
    pipe.prompter = FixedPrompter(inp)
    return g, pipe, image, control


def test_pipeline_call_matches_the_reference_call_end_to_end():
    """`pipe(prompt, negative_prompt, input_image, control_signal_video, seed=0, height=64, width=96, num_frames=9,
    num_inference_steps=3, cfg_scale=5.0, tiled=True, tile_size=(6,8), tile_stride=(3,4), controlnet=True)` — the product against
    what the reference's own `__call__` returned for exactly these arguments:
      * the same forwards in the same order: (expert 2?, ControlNet 2?, bf16-rounded timestep) per model_fn call — bit-exact,
      * the prompter is asked the way the reference's unit asks it,
      * final latents: no further from the reference's fp32 run than the reference's bf16 run is (x 1.25), and within 2 x that of
        the bf16 run itself (CFG x 5 on a random tiny model amplifies every rounding: the reference's own bf16 run sits 0.16 away),
      * 9 PIL frames of 96 x 64; PSNR against the fp32 run's frames within 1 dB of the reference bf16 run's."""
    from PIL import Image
    g, pipe, image, control = _pipe()
    calls = []
    real_fn = pipe.model_fn

    def spy(**kw):
        calls.append((float(kw["dit"] is pipe.dit2), float(kw.get("controlnet") is pipe.controlnet2), float(kw["timestep"].float())))
        return real_fn(**kw)
    pipe.model_fn = spy           # GF:161: the reference's own swap point
    seen = {}
    real_decode = pipe.vae.decode

    def spy_decode(hidden_states, *a, **k):
        seen["latents"] = hidden_states.detach().clone()
        return real_decode(hidden_states, *a, **k)
    pipe.vae.decode = spy_decode
    frames = pipe(prompt=gi.PIPELINE_PROMPTS[0], negative_prompt=gi.PIPELINE_PROMPTS[1], input_image=image,
                  control_signal_video=control, **gi.PIPELINE_CALL_KWARGS)
    assert isinstance(frames, list) and len(frames) == 9 and all(isinstance(f, Image.Image) and f.size == (96, 64) for f in frames)
    assert pipe.prompter.calls == [(gi.PIPELINE_PROMPTS[0], None), (gi.PIPELINE_PROMPTS[1], None)]
    want_calls = [tuple(r) for r in g["model_fn_calls_bf16"].tolist()]
    assert calls == want_calls, (calls, want_calls)
    lat = seen["latents"].float().cpu()
    f32 = torch.from_numpy(g["latents_f32"])
    ref_bf = gi.from_u16(g["latents_bf16"]).float()
    e, e_ref, e_bf = rel_l2(lat, f32), rel_l2(ref_bf, f32), rel_l2(lat, ref_bf)
    assert tuple(lat.shape) == (1, 16, 3, 8, 12)
    assert e < 1.25 * e_ref + 1e-3 and e_bf < 2 * e_ref, f"latents vs fp32 {e:.3e}, vs ref-bf16 {e_bf:.3e} (reference bf16 vs fp32 {e_ref:.3e})"
    got_u8 = np.stack([np.array(f) for f in frames])
    p, p_ref = psnr_u8(got_u8, g["frames_u8_f32"]), psnr_u8(g["frames_u8_bf16"], g["frames_u8_f32"])
    assert got_u8.shape == g["frames_u8_bf16"].shape == (9, 64, 96, 3)
    assert p > p_ref - 1.0, f"frames PSNR vs the fp32 run {p:.2f} dB (reference bf16 run {p_ref:.2f} dB)"
    print(f"g13: latents vs fp32 {e:.3e} (reference bf16 {e_ref:.3e}), vs ref-bf16 {e_bf:.3e}; frames PSNR {p:.2f} dB (reference {p_ref:.2f} dB)")


def test_post_loop_chain_on_the_reference_latents():
    """The chaotic loop taken out: the product's tiled decode + frame conversion (GF:733-735) applied to the REFERENCE's final
    bf16 latents against the reference's own decoded video / uint8 frames of that run — two bf16 evaluations of the same VAE."""
    g, pipe, _, _ = _pipe()
    z = gi.from_u16(g["latents_bf16"]).cuda()
    kw = gi.PIPELINE_CALL_KWARGS
    video = pipe.vae.decode(z, device="cuda", tiled=True, tile_size=kw["tile_size"], tile_stride=kw["tile_stride"])
    ref_video = gi.from_u16(g["video_bf16"]).float()
    e = rel_l2(video.float().cpu(), ref_video)
    assert tuple(video.shape) == (1, 3, 9, 64, 96) and e < 2e-2, f"decode of the reference latents: rel-L2 {e:.3e}"
    u8 = np.stack([np.array(f) for f in pipe.vae_output_to_video(video)])
    diff = np.abs(u8.astype(np.int32) - g["frames_u8_bf16"].astype(np.int32))
    p = psnr_u8(u8, g["frames_u8_bf16"])
    assert p > 38.0 and float((diff <= 2).mean()) > 0.97, f"frames vs the reference's: PSNR {p:.2f} dB, within 2 levels {float((diff <= 2).mean()):.4f}"
    # the conversion itself is bit-exact on the reference's own video tensor (UTIL:76-91)
    exact = np.stack([np.array(f) for f in pipe.vae_output_to_video(gi.from_u16(g["video_bf16"]).cuda())])
    assert np.array_equal(exact, g["frames_u8_bf16"])
