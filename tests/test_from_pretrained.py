"""`WanVideoPipeline.from_pretrained` as the reference's inference script calls it (INF:81-106): four ModelConfigs with
local paths — two DiT experts as lists of .safetensors shards, the umT5 encoder `.pth`, the VAE `.pth` — plus a
tokenizer_config, controlnet=True.  The checkpoints here are synthetic but written under the reference's key names
(WanModel DIT:423-466 via gen_inputs.dit_sd; T5 via gen_inputs.t5_sd; VAE = the reference's own parameter list from
g6_vae.npz, stored like Wan2.1_VAE.pth WITHOUT the 'model.' prefix); every load is strict."""
import json
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN

BF = torch.bfloat16
torch.set_grad_enabled(False)
T5_TINY = dict(vocab=64, dim=64, dim_attn=256, dim_ffn=128, num_heads=4, num_layers=2, num_buckets=32)   # dim = TINY text_dim


def _write_tokenizer(path):
    from tokenizers import Tokenizer, models, pre_tokenizers
    os.makedirs(path)
    vocab = {"<pad>": 0, "</s>": 1, "<unk>": 2}
    for i, w in enumerate("the pendulum swings striking and toppling red block a ball rolls".split()):
        vocab[w] = 3 + i
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "pad_token": "<pad>", "eos_token": "</s>",
                   "unk_token": "<unk>", "model_max_length": 512}, f)


def _write_checkpoints(root):
    from safetensors.torch import save_file
    paths = {}
    for name, seed in (("high_noise_model", 41), ("low_noise_model", 43)):
        sd = gi.dit_sd(gi.TINY, seed=seed)
        keys = sorted(sd)
        os.makedirs(root / name)
        shards = []
        for i in range(3):                      # sharded like diffusion_pytorch_model-0000i-of-00003.safetensors
            f = str(root / name / f"diffusion_pytorch_model-{i + 1:05d}-of-00003.safetensors")
            save_file({k: sd[k].contiguous() for k in keys[i::3]}, f)
            shards.append(f)
        paths[name] = (shards, sd)
    t5 = gi.t5_sd(T5_TINY, seed=7)
    torch.save(t5, str(root / "models_t5_umt5-xxl-enc-bf16.pth"))
    g6 = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    vae = {str(k): v for k, v in gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61).items()}
    # ^ keys without 'model.', as in Wan2.1_VAE.pth
    torch.save(vae, str(root / "Wan2.1_VAE.pth"))
    _write_tokenizer(str(root / "google" / "umt5-xxl"))
    return paths, t5, vae


def test_from_pretrained_runs_the_reference_call(tmp_path):
    from goal_force_amd.pipeline import ModelConfig, WanVideoPipeline
    from goal_force_amd.text_encoder import WanTextEncoder
    from goal_force_amd.vae import WanVideoVAE
    paths, t5, vae = _write_checkpoints(tmp_path)
    pipe = WanVideoPipeline.from_pretrained(
        torch_dtype=BF, device="cpu",
        tokenizer_config=ModelConfig(model_id="Wan-AI/Wan2.1-T2V-1.3B", origin_file_pattern="google/*",
                                     path=str(tmp_path / "google" / "umt5-xxl")),
        model_configs=[
            ModelConfig(model_id="Wan-AI/Wan2.2-I2V-A14B", origin_file_pattern="high_noise_model/diffusion_pytorch_model*.safetensors",
                        offload_device=None, path=paths["high_noise_model"][0]),
            ModelConfig(model_id="Wan-AI/Wan2.2-I2V-A14B", origin_file_pattern="low_noise_model/diffusion_pytorch_model*.safetensors",
                        offload_device=None, path=paths["low_noise_model"][0]),
            ModelConfig(model_id="Wan-AI/Wan2.2-I2V-A14B", origin_file_pattern="models_t5_umt5-xxl-enc-bf16.pth",
                        offload_device=None, path=str(tmp_path / "models_t5_umt5-xxl-enc-bf16.pth")),
            ModelConfig(model_id="Wan-AI/Wan2.2-I2V-A14B", origin_file_pattern="Wan2.1_VAE.pth", offload_device=None,
                        path=str(tmp_path / "Wan2.1_VAE.pth")),
        ],
        controlnet=True, controlnet_num_layers=1)
    # the two experts, in the order given (GF:529-533), sized from the files
    for m, name in ((pipe.dit, "high_noise_model"), (pipe.dit2, "low_noise_model")):
        want = paths[name][1]
        got = m.state_dict()
        assert sorted(got) == sorted(want) and all(torch.equal(got[k], want[k]) for k in want)
        assert (m.dim, m.in_dim, m.num_heads, len(m.blocks)) == (256, 36, 2, 2)
    assert isinstance(pipe.text_encoder, WanTextEncoder) and pipe.text_encoder.num_layers == 2
    assert all(torch.equal(v, t5[k]) for k, v in pipe.text_encoder.state_dict().items())
    assert pipe.prompter.text_encoder is pipe.text_encoder                           # GF:585
    assert isinstance(pipe.vae, WanVideoVAE)
    assert all(torch.equal(v, vae[k[len("model."):]]) for k, v in pipe.vae.state_dict().items())   # VAE:1256-1262
    assert pipe.height_division_factor == pipe.width_division_factor == 16           # GF:580-582
    # tokenizer_config.path reached the prompter (GF:586): 512-token right-padded ids + mask
    ids, mask = pipe.prompter.tokenize("the  red ball   rolls")
    assert tuple(ids.shape) == (1, 512) and int(mask.sum()) == 4 and ids[0, :4].tolist() == [3, 9, 12, 13]
    # ControlNet blocks = copies of expert blocks 0..N-1; controlnet2 from the low-noise expert (GF:559-568)
    for cn, dit in ((pipe.controlnet, pipe.dit), (pipe.controlnet2, pipe.dit2)):
        a, b = cn.controlnet_dit.blocks[0].state_dict(), dit.blocks[0].state_dict()
        assert all(torch.equal(a[k], b[k]) for k in b)
        assert all(float(p.abs().max()) == 0 for c in cn.controlnet_zero_convs_after for p in c.parameters())
    # load_controlnet_weights: 'pipe.controlnet.' prefix stripped, strict (GF:176-178)
    from safetensors.torch import save_file
    csd = gi.controlnet_sd(gi.TINY, 1, seed=42)
    save_file({"pipe.controlnet." + k: v.contiguous() for k, v in csd.items()}, str(tmp_path / "step-10.safetensors"))
    pipe.load_controlnet_weights(pipe.controlnet, str(tmp_path / "step-10.safetensors"), torch_dtype=BF)
    assert all(torch.equal(v, csd[k]) for k, v in pipe.controlnet.state_dict().items())
    assert all(float(p.abs().max()) == 0 for c in pipe.controlnet2.controlnet_zero_convs_after for p in c.parameters())
    pipe.enable_vram_management()                                                    # INF:111


def test_from_pretrained_model_id_resolution_and_refusals(tmp_path):
    from goal_force_amd._lib import GoalForceError
    from goal_force_amd.pipeline import ModelConfig, WanVideoPipeline
    paths, t5, vae = _write_checkpoints(tmp_path / "models" / "Wan-AI" / "X")
    # model_id + origin_file_pattern resolve under local_model_path (what the reference's download would have produced)
    mc = ModelConfig(model_id="Wan-AI/X", origin_file_pattern="high_noise_model/diffusion_pytorch_model*.safetensors",
                     local_model_path=str(tmp_path / "models"))
    pipe = WanVideoPipeline.from_pretrained(device="cpu", model_configs=[mc])
    assert pipe.dit is not None and pipe.dit2 is None and len(mc.path) == 3
    # redirect_common_files (GF:498-510): the encoder / VAE named under another Wan repository are looked up where the reference's download puts them
    _write_checkpoints(tmp_path / "models" / "Wan-AI" / "Wan2.1-T2V-1.3B")
    cfgs = [ModelConfig(model_id="Wan-AI/Wan2.2-I2V-A14B", origin_file_pattern="Wan2.1_VAE.pth", local_model_path=str(tmp_path / "models")),
            ModelConfig(model_id="Wan-AI/Wan2.2-I2V-A14B", origin_file_pattern="models_t5_umt5-xxl-enc-bf16.pth", local_model_path=str(tmp_path / "models"))]
    pipe = WanVideoPipeline.from_pretrained(device="cpu", model_configs=cfgs)
    assert pipe.vae is not None and pipe.text_encoder is not None and cfgs[0].model_id == "Wan-AI/Wan2.1-T2V-1.3B"
    assert cfgs[0].path == str(tmp_path / "models" / "Wan-AI" / "Wan2.1-T2V-1.3B" / "Wan2.1_VAE.pth")
    # the DEFAULT tokenizer_config (GF:486: Wan2.1-T2V-1.3B, "google/*" under ./models) — what train.py:52-55 relies on
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        pipe = WanVideoPipeline.from_pretrained(device="cpu", model_configs=[ModelConfig(path=str(tmp_path / "models" / "Wan-AI" / "Wan2.1-T2V-1.3B" / "Wan2.1_VAE.pth"))])
        assert pipe.prompter.tokenizer is not None and int(pipe.prompter.tokenize("the red ball")[1].sum()) == 3
    finally:
        os.chdir(cwd)
    with pytest.raises(GoalForceError, match="no network"):       # ... and with the redirection off the file is not where the config says
        WanVideoPipeline.from_pretrained(device="cpu", redirect_common_files=False, model_configs=[
            ModelConfig(model_id="Wan-AI/Wan2.2-I2V-A14B", origin_file_pattern="Wan2.1_VAE.pth", local_model_path=str(tmp_path / "models"))])
    with pytest.raises(GoalForceError, match="no network"):
        WanVideoPipeline.from_pretrained(device="cpu", model_configs=[ModelConfig(model_id="Wan-AI/absent",
                                                                                   origin_file_pattern="*.pth",
                                                                                   local_model_path=str(tmp_path / "models"))])
    torch.save({"something.else": torch.zeros(3)}, str(tmp_path / "other.pth"))
    with pytest.raises(NotImplementedError, match="out of scope"):
        WanVideoPipeline.from_pretrained(device="cpu", model_configs=[ModelConfig(path=str(tmp_path / "other.pth"))])
    with pytest.raises(GoalForceError, match="needs a DiT expert"):
        WanVideoPipeline.from_pretrained(device="cpu", model_configs=[], controlnet=True, controlnet_num_layers=1)


@pytest.mark.gpu
def test_pipeline_from_checkpoints_end_to_end_tiny(tmp_path):
    """The whole reference call sequence on the GPU at tiny size: from_pretrained (4 ModelConfigs + tokenizer) ->
    load_controlnet_weights -> pipe(prompt=..., input_image=..., control_signal_video=...) -> 9 PIL frames; the denoised
    latents equal a run fed with the separately computed pre-loop tensors."""
    from PIL import Image
    from safetensors.torch import save_file
    from goal_force_amd.pipeline import ModelConfig, WanVideoPipeline
    paths, t5, vae = _write_checkpoints(tmp_path)
    cfgs = [ModelConfig(path=paths["high_noise_model"][0]), ModelConfig(path=paths["low_noise_model"][0]),
            ModelConfig(path=str(tmp_path / "models_t5_umt5-xxl-enc-bf16.pth")), ModelConfig(path=str(tmp_path / "Wan2.1_VAE.pth"))]
    pipe = WanVideoPipeline.from_pretrained(torch_dtype=BF, device="cuda", model_configs=cfgs, controlnet=True,
                                            controlnet_num_layers=1,
                                            tokenizer_config=ModelConfig(path=str(tmp_path / "google" / "umt5-xxl")))
    csd = gi.controlnet_sd(gi.TINY, 1, seed=42)
    save_file({"pipe.controlnet." + k: v.contiguous() for k, v in csd.items()}, str(tmp_path / "step-10.safetensors"))
    pipe.load_controlnet_weights(pipe.controlnet, str(tmp_path / "step-10.safetensors"))
    image, control = gi.preloop_inputs()
    kw = dict(prompt="the pendulum swings", negative_prompt="a red block", input_image=image, num_frames=9, height=64,
              width=96, seed=3, tiled=False, controlnet=True, control_signal_video=control, num_inference_steps=3)
    frames = pipe(**kw)
    assert len(frames) == 9 and frames[0].size == (96, 64) and isinstance(frames[0], Image.Image)
    lat = pipe(**kw, output_type="latent")
    assert tuple(lat.shape) == (1, 16, 3, 8, 12) and bool(torch.isfinite(lat.float()).all())
    ctx_p = pipe.prompter.encode_prompt("the pendulum swings", device="cuda")
    ctx_n = pipe.prompter.encode_prompt("a red block", positive=False, device="cuda")
    assert tuple(ctx_p.shape) == (1, 512, 64) and float(ctx_p[:, 3:].abs().max()) == 0      # zeroed past the prompt
    y = pipe.embed_image(image, 9, 64, 96, False, (30, 52), (15, 26))
    cl = pipe.embed_control_video(control, False, (30, 52), (15, 26))
    lat2 = pipe(context_posi=ctx_p, context_nega=ctx_n, y=y, control_signal_video_latents=cl, num_frames=9, height=64,
                width=96, seed=3, controlnet=True, num_inference_steps=3, output_type="latent")
    assert torch.equal(lat, lat2)


def test_load_model_adopts_checkpoint_tensors_without_a_second_copy(tmp_path):
    """checkpoints.load_model builds the module with its parameters on the meta device (no storage, no random init) and takes
    over the tensors read from the files (`load_state_dict(assign=True)`), each cast to torch_dtype as it is read — not a random
    fp32 model that the checkpoint is then copied into (for an A14B expert: 57 GB and minutes of CPU RNG).  fp32 shards in,
    bf16 parameters out; a file that lacks a parameter fails by name."""
    from safetensors.torch import save_file
    from goal_force_amd import checkpoints as ck
    from goal_force_amd.dit import WanModel
    with ck.params_on_meta():
        m = WanModel(has_image_input=False, require_clip_embedding=False, **gi.TINY)
    assert all(p.is_meta for p in m.parameters()) and not any(f.is_meta for f in m.freqs), "parameters on meta, RoPE tables real"
    sd = {k: v.float() for k, v in gi.dit_sd(gi.TINY, seed=41).items()}
    path = os.path.join(tmp_path, "dit.safetensors")
    save_file(sd, path)
    seen = {}
    real = ck.load_state_dict

    def spy(*a, **kw):
        seen["sd"] = real(*a, **kw)
        return seen["sd"]

    ck.load_state_dict = spy
    try:
        kind, mod = ck.load_model(ck.ModelConfig(path=path), torch_dtype=BF, device="cpu")
    finally:
        ck.load_state_dict = real
    assert kind == "wan_video_dit" and all(p.dtype == BF and not p.is_meta for p in mod.parameters())
    w = mod.blocks[0].ffn[0].weight
    assert w.data_ptr() == seen["sd"]["blocks.0.ffn.0.weight"].data_ptr(), "the parameter IS the tensor read from the file"
    assert torch.equal(w, sd["blocks.0.ffn.0.weight"].to(BF))
    del sd["blocks.1.ffn.2.bias"]
    save_file(sd, path)
    with pytest.raises(RuntimeError, match="blocks.1.ffn.2.bias"):
        ck.load_model(ck.ModelConfig(path=path), torch_dtype=BF, device="cpu")


def test_params_on_meta_is_nested_safe_and_inert_for_other_threads():
    """the meta-device construction patches torch.nn.Module.register_parameter process-wide while active — nested
    use must restore the original exactly once, and a module another thread builds meanwhile must get REAL parameters."""
    import threading
    import torch
    from goal_force_amd.checkpoints import params_on_meta
    orig = torch.nn.Module.register_parameter
    other = {}
    with params_on_meta():
        with params_on_meta():
            inner = torch.nn.Linear(4, 4)
        mid = torch.nn.Linear(4, 4)                 # still inside the outer context: still meta
        t = threading.Thread(target=lambda: other.setdefault("m", torch.nn.Linear(4, 4)))
        t.start()
        t.join()
    after = torch.nn.Linear(4, 4)
    assert inner.weight.is_meta and mid.weight.is_meta
    assert not other["m"].weight.is_meta and not after.weight.is_meta
    assert torch.nn.Module.register_parameter is orig
