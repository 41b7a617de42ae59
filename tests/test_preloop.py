"""Pre-loop conditioning units (GF:751-763 noise, GF:791-805 control-video embedder, GF:887-917 image
embedder) against tests/golden/g10_preloop.npz — outputs of the REFERENCE's own unit classes (make_goldens.py::g10_preloop).

CPU: oracle/preloop_oracle.py bit-exact in bf16 (same torch ops).  GPU: `WanVideoPipeline.embed_image`,
`.embed_control_video`, `.generate_noise`, `.check_resize_height_width` through the HIP VAE encoder: mask channels,
noise and shapes bit-exact; latent channels rel-L2 vs the fp32 golden <= max(1.5e-2, 1.5x the reference-bf16's own
distance from fp32) (the encoder is ~40 bf16 convs deep, same bar as tests/test_vae.py)."""
import hashlib
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN, rel_l2
from oracle import preloop_oracle as po

BF = torch.bfloat16
torch.set_grad_enabled(False)
H, W, F = 64, 96, 9
TILE = dict(tile_size=(6, 8), tile_stride=(3, 4))


def _fixture():
    g = np.load(os.path.join(GOLDEN, "g10_preloop.npz"))
    g6 = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    sd = gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61)
    image, control = gi.preloop_inputs()
    assert gi.same_checksum(gi.checksum([torch.from_numpy(np.array(image)).float(), control]), g["ck_inputs"])
    return g, sd, image, control


def test_oracle_noise_and_shape_check_match_reference():
    g = np.load(os.path.join(GOLDEN, "g10_preloop.npz"))
    assert torch.equal(po.noise(H, W, F, 0), gi.from_u16(g["noise_bf16"]))
    assert torch.equal(po.noise(H, W, F, 0, dtype=torch.float32), torch.from_numpy(g["noise_f32"]))
    for s in range(4):           # BASELINE configs 2/3: seeds 0..3 at [1,16,21,60,104]
        n = po.noise(480, 832, 81, s)
        assert tuple(n.shape) == (1, 16, 21, 60, 104)
        assert hashlib.sha256(gi.to_u16(n).tobytes()).hexdigest() == str(g["noise_full_sha256"][s])
    got = [po.shape_check(*s) for s in gi.PRELOOP_SHAPES]
    assert np.array_equal(np.array(got), g["shape_check"])


@pytest.mark.parametrize("tiled", [False, True])
def test_oracle_embedders_match_reference(tiled):
    g, sd, image, control = _fixture()
    tag = "tiled" if tiled else "plain"
    y = po.image_y(image, F, H, W, sd, tiled=tiled, **TILE)
    assert torch.equal(y, gi.from_u16(g[f"y_{tag}_bf16"]))                      # mask + latents, bit-exact
    c = po.control_latents(control, sd, tiled=tiled, **TILE)
    assert torch.equal(c, gi.from_u16(g[f"control_{tag}_bf16"]))
    if not tiled:
        sd32 = {k: v.float() for k, v in sd.items()}
        y32 = po.image_y(image, F, H, W, sd32, dtype=torch.float32, tiled=False)
        assert rel_l2(y32, torch.from_numpy(g["y_plain_f32"])) < 1e-5
        assert torch.equal(y32[:, :4], torch.from_numpy(g["y_plain_f32"])[:, :4])


def test_mask_channels_are_first_frame_only():
    g = np.load(os.path.join(GOLDEN, "g10_preloop.npz"))
    m = gi.from_u16(g["y_plain_bf16"])[0, :4]
    assert bool((m[:, 0] == 1).all()) and bool((m[:, 1:] == 0).all())


def _gpu_pipe(sd):
    from goal_force_amd.pipeline import WanVideoPipeline
    from goal_force_amd.vae import WanVideoVAE
    v = WanVideoVAE()
    v.load_state_dict({"model." + k: t for k, t in sd.items()}, strict=True)
    return WanVideoPipeline.from_modules(None, vae=v.to(BF).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("tiled", [False, True])
def test_hip_embedders_vs_reference_golden(tiled):
    g, sd, image, control = _fixture()
    pipe = _gpu_pipe(sd)
    tag = "tiled" if tiled else "plain"
    y = pipe.embed_image(image, F, H, W, tiled, **TILE).cpu()
    ref_bf, ref32 = gi.from_u16(g[f"y_{tag}_bf16"]), torch.from_numpy(g[f"y_{tag}_f32"])
    assert y.dtype == BF and tuple(y.shape) == tuple(ref_bf.shape) == (1, 20, 3, 8, 12)
    assert torch.equal(y[:, :4], ref_bf[:, :4])                                 # mask channels: bit-exact
    e, e_ref = rel_l2(y[:, 4:].float(), ref32[:, 4:]), rel_l2(ref_bf[:, 4:].float(), ref32[:, 4:])
    assert e < max(1.5e-2, 1.5 * e_ref), f"y latents: vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
    c = pipe.embed_control_video(control.cuda(), tiled, **TILE).cpu()
    ref_bf, ref32 = gi.from_u16(g[f"control_{tag}_bf16"]), torch.from_numpy(g[f"control_{tag}_f32"])
    assert c.dtype == BF and tuple(c.shape) == tuple(ref_bf.shape)
    e, e_ref = rel_l2(c.float(), ref32), rel_l2(ref_bf.float(), ref32)
    assert e < max(1.5e-2, 1.5 * e_ref), f"control latents: vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
    # the host control video (what the dataset hands over, DS:889) gives the same latents as a device one
    assert torch.equal(pipe.embed_control_video(control, tiled, **TILE).cpu(), c)


@pytest.mark.gpu
def test_hip_pipeline_noise_and_shapes_bit_exact():
    g, sd, image, control = _fixture()
    pipe = _gpu_pipe(sd)
    n = pipe.generate_noise((1, 16, 3, 8, 12), seed=0)
    assert n.is_cuda and torch.equal(n.cpu(), gi.from_u16(g["noise_bf16"]))
    for s in range(4):
        n = pipe.generate_noise((1, 16, 21, 60, 104), seed=s, rand_device="cpu")
        assert hashlib.sha256(gi.to_u16(n.cpu()).tobytes()).hexdigest() == str(g["noise_full_sha256"][s])
    got = [pipe.check_resize_height_width(*s) for s in gi.PRELOOP_SHAPES]
    assert np.array_equal(np.array(got), g["shape_check"])
    assert pipe.height_division_factor == pipe.vae.upsampling_factor * 2 == 16          # GF:580-582


@pytest.mark.gpu
def test_call_requires_image_conditioning_when_the_dit_needs_it():
    """GF:896: the I2V expert (require_vae_embedding) cannot run without `y`; the reference fails on the channel count
    of the patch embedding, ours must not silently treat the missing channels as zeros."""
    from goal_force_amd._lib import GoalForceError
    from goal_force_amd.dit import WanModel
    from goal_force_amd.pipeline import WanVideoPipeline
    cfg = dict(gi.TINY)
    dit = WanModel(has_image_input=False, require_clip_embedding=False, **cfg)
    dit.load_state_dict(gi.dit_sd(cfg, seed=41), strict=True)
    pipe = WanVideoPipeline.from_modules(dit.to(BF).cuda())
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    with pytest.raises(GoalForceError, match="input_image"):
        pipe(context_posi=inp["ctx_posi"], context_nega=inp["ctx_nega"], height=64, width=96, num_frames=9,
             num_inference_steps=1, output_type="latent")


def test_frames_to_uint8_matches_reference_bytes():
    """Post-loop host step (GF:735 -> UTIL:76-91) on CPU tensors: the bf16 arithmetic rounds before the uint8 cast."""
    from goal_force_amd.pipeline import WanVideoPipeline
    g = np.load(os.path.join(GOLDEN, "g10_preloop.npz"))
    pipe = WanVideoPipeline(device="cpu")
    vid = gi.preloop_decoded_video()
    assert np.array_equal(pipe.frames_uint8(vid).numpy(), g["frames_u8_bf16"])
    assert np.array_equal(pipe.frames_uint8(vid.float()).numpy(), g["frames_u8_f32"])
    assert not np.array_equal(g["frames_u8_bf16"], g["frames_u8_f32"])      # the dtype of the arithmetic matters
    ims = pipe.vae_output_to_video(vid)
    assert len(ims) == 2 and ims[0].size == (8, 8) and np.array_equal(np.array(ims[1]), g["frames_u8_bf16"][1])


# ------------------------------------------------------------------ training-side caller: forward_preprocess (train.py:76-118; g16)
def _g16():
    g = np.load(os.path.join(GOLDEN, "g16_forward_preprocess.npz"))
    g6 = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    sd = gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61)
    clip, (_, control), inp = gi.training_clip(), gi.preloop_inputs(), gi.tiny_inputs()
    assert gi.same_checksum(gi.checksum([torch.from_numpy(np.stack([np.array(f) for f in clip])).float(), control, inp["ctx_posi"]]), g["ck_inputs"])
    return g, sd, clip, control, inp


def test_oracle_training_inputs_match_the_reference_units_in_training_mode():
    """The reference's units, run in training mode the way WanTrainingModule.forward_preprocess runs them (g16): the clip's latents
    (InputVideoEmbedder), the control latents and y — oracle bit-exact in bf16; the tensors forward_preprocess hands to training_loss."""
    g, sd, clip, control, inp = _g16()
    # every tensor the reference's units left in the inputs (control_signal_video = the raw clip that went in; latents IS noise in training mode)
    assert [str(k) for k in g["keys"]] == ["context", "control_signal_video", "control_signal_video_latents", "input_latents", "latents", "noise", "y"]
    assert torch.equal(po.input_latents(clip, sd), gi.from_u16(g["input_latents_bf16"]))
    assert torch.equal(po.control_latents(control, sd, tiled=False), gi.from_u16(g["control_signal_video_latents_bf16"]))
    assert torch.equal(po.image_y(clip[0], F, H, W, sd, tiled=False), gi.from_u16(g["y_bf16"]))
    assert torch.equal(inp["ctx_posi"], gi.from_u16(g["context_bf16"]))
    sd32 = {k: v.float() for k, v in sd.items()}
    assert rel_l2(po.input_latents(clip, sd32, dtype=torch.float32), torch.from_numpy(g["input_latents_f32"])) < 1e-5


@pytest.mark.gpu
def test_hip_forward_preprocess_vs_reference_golden():
    """training.forward_preprocess(pipe, item) -> the keyword arguments of training_loss, against g16."""
    from goal_force_amd import training as tr
    from goal_force_amd._lib import GoalForceError
    g, sd, clip, control, inp = _g16()
    pipe = _gpu_pipe(sd)

    class Prompter:
        calls = []

        def encode_prompt(self, prompt, positive=True, device="cuda"):
            self.calls.append((prompt, positive))
            return inp["ctx_posi"].to(device)
    pipe.prompter = Prompter()
    item = {"video": clip, "prompt": gi.PIPELINE_PROMPTS[0], "control_video": control}
    with pytest.raises(GoalForceError, match="training mode"):
        tr.forward_preprocess(pipe, item)
    pipe.scheduler.set_timesteps(1000, training=True)
    out = tr.forward_preprocess(pipe, item)
    assert sorted(out) == ["context", "control_signal_video_latents", "input_latents", "noise", "y"]
    assert Prompter.calls == [(gi.PIPELINE_PROMPTS[0], None)]
    assert tuple(out["noise"].shape) == (1, 16, 3, 8, 12) and out["noise"].dtype == BF and out["noise"].is_cuda
    assert torch.equal(out["context"].cpu(), gi.from_u16(g["context_bf16"]))
    assert torch.equal(out["y"][:, :4].cpu(), gi.from_u16(g["y_bf16"])[:, :4])
    for k, got in (("input_latents", out["input_latents"]), ("control_signal_video_latents", out["control_signal_video_latents"]), ("y", out["y"])):
        ref_bf, ref32 = gi.from_u16(g[f"{k}_bf16"]).float(), torch.from_numpy(g[f"{k}_f32"])
        sl = slice(4, None) if k == "y" else slice(None)
        e, e_ref = rel_l2(got.cpu().float()[:, sl], ref32[:, sl]), rel_l2(ref_bf[:, sl], ref32[:, sl])
        assert got.dtype == BF and tuple(got.shape) == tuple(ref32.shape)
        assert e < max(1.5e-2, 1.5 * e_ref), f"{k}: vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
