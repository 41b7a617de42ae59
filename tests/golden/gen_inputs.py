"""Seeded inputs + random weights shared by the golden generator (make_goldens.py, build container
only) and the tests.  Pure torch CPU RNG: the same torch build on the GPU box reproduces the same
tensors; every fixture stores a checksum of what was generated so a drift is caught, not hidden.

State-dict key names are the reference's (WanModel: diffsynth/models/wan_video_dit.py:273-340;
ControlNet checkpoint layout: SURVEY.md §5 'checkpoint / resume').
"""
from __future__ import annotations

import math

import numpy as np
import torch

TINY = dict(dim=256, in_dim=36, ffn_dim=512, out_dim=16, text_dim=64, freq_dim=256, eps=1e-6,
            patch_size=(1, 2, 2), num_heads=2, num_layers=2)
TINY_LATENT = (1, 16, 3, 8, 12)       # -> grid (3,4,6) = 72 tokens
TINY_CTX_LEN = 7
TINY_CONTROLNET_LAYERS = 1

# Wan-1.3B-like mid size (head_dim 128, 12 heads)
MID = dict(dim=1536, in_dim=36, ffn_dim=8960, out_dim=16, text_dim=4096, freq_dim=256, eps=1e-6,
           patch_size=(1, 2, 2), num_heads=12, num_layers=1)

A14B = dict(dim=5120, in_dim=36, ffn_dim=13824, out_dim=16, text_dim=4096, freq_dim=256, eps=1e-6,
            patch_size=(1, 2, 2), num_heads=40, num_layers=40)


def _randn(g, shape, std=1.0):
    return torch.randn(shape, generator=g) * std


def block_sd(g, dim, ffn_dim, prefix, dtype):
    sd = {}
    for att in ("self_attn", "cross_attn"):
        for p in ("q", "k", "v", "o"):
            sd[f"{prefix}{att}.{p}.weight"] = _randn(g, (dim, dim), 1.0 / math.sqrt(dim))
            sd[f"{prefix}{att}.{p}.bias"] = _randn(g, (dim,), 0.02)
        sd[f"{prefix}{att}.norm_q.weight"] = 1.0 + _randn(g, (dim,), 0.1)
        sd[f"{prefix}{att}.norm_k.weight"] = 1.0 + _randn(g, (dim,), 0.1)
    sd[f"{prefix}norm3.weight"] = 1.0 + _randn(g, (dim,), 0.1)
    sd[f"{prefix}norm3.bias"] = _randn(g, (dim,), 0.05)
    sd[f"{prefix}ffn.0.weight"] = _randn(g, (ffn_dim, dim), 1.0 / math.sqrt(dim))
    sd[f"{prefix}ffn.0.bias"] = _randn(g, (ffn_dim,), 0.02)
    sd[f"{prefix}ffn.2.weight"] = _randn(g, (dim, ffn_dim), 1.0 / math.sqrt(ffn_dim))
    sd[f"{prefix}ffn.2.bias"] = _randn(g, (dim,), 0.02)
    sd[f"{prefix}modulation"] = _randn(g, (1, 6, dim), 1.0 / math.sqrt(dim))
    return {k: v.to(dtype) for k, v in sd.items()}


def dit_sd(cfg, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    d, pk = cfg["dim"], cfg["in_dim"] * 4
    sd = {
        "patch_embedding.weight": _randn(g, (d, cfg["in_dim"], 1, 2, 2), 1.0 / math.sqrt(pk)),
        "patch_embedding.bias": _randn(g, (d,), 0.02),
        "text_embedding.0.weight": _randn(g, (d, cfg["text_dim"]), 1.0 / math.sqrt(cfg["text_dim"])),
        "text_embedding.0.bias": _randn(g, (d,), 0.02),
        "text_embedding.2.weight": _randn(g, (d, d), 1.0 / math.sqrt(d)),
        "text_embedding.2.bias": _randn(g, (d,), 0.02),
        "time_embedding.0.weight": _randn(g, (d, cfg["freq_dim"]), 1.0 / math.sqrt(cfg["freq_dim"])),
        "time_embedding.0.bias": _randn(g, (d,), 0.02),
        "time_embedding.2.weight": _randn(g, (d, d), 1.0 / math.sqrt(d)),
        "time_embedding.2.bias": _randn(g, (d,), 0.02),
        "time_projection.1.weight": _randn(g, (6 * d, d), 1.0 / math.sqrt(d)),
        "time_projection.1.bias": _randn(g, (6 * d,), 0.02),
        "head.head.weight": _randn(g, (cfg["out_dim"] * 4, d), 1.0 / math.sqrt(d)),
        "head.head.bias": _randn(g, (cfg["out_dim"] * 4,), 0.02),
        "head.modulation": _randn(g, (1, 2, d), 1.0 / math.sqrt(d)),
    }
    sd = {k: v.to(dtype) for k, v in sd.items()}
    for i in range(cfg["num_layers"]):
        sd.update(block_sd(g, d, cfg["ffn_dim"], f"blocks.{i}.", dtype))
    return sd


def controlnet_sd(cfg, n_layers, seed, dtype=torch.bfloat16, zero_convs_zero=False):
    g = torch.Generator().manual_seed(seed)
    d = cfg["dim"]
    sd = {
        "controlnet_patch_embedding.patch_embedding.weight": _randn(g, (d, 16, 1, 2, 2), 1.0 / 8.0).to(dtype),
        "controlnet_patch_embedding.patch_embedding.bias": _randn(g, (d,), 0.02).to(dtype),
    }
    for i in range(n_layers):
        sd.update(block_sd(g, d, cfg["ffn_dim"], f"controlnet_dit.blocks.{i}.", dtype))
    for i in range(n_layers):
        w = _randn(g, (d, d, 1), 0.5 / math.sqrt(d))
        b = _randn(g, (d,), 0.02)
        if zero_convs_zero:
            w, b = torch.zeros_like(w), torch.zeros_like(b)
        sd[f"controlnet_zero_convs_after.{i}.weight"] = w.to(dtype)
        sd[f"controlnet_zero_convs_after.{i}.bias"] = b.to(dtype)
    return sd


def tiny_inputs(seed=1234, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    b, c, f, h, w = TINY_LATENT
    latents = _randn(g, (b, c, f, h, w)).to(dtype)
    y = _randn(g, (b, 20, f, h, w)).to(dtype)
    y[:, :4] = (y[:, :4] > 0).to(dtype)  # mask channels are {0,1} (GF:905-912)
    control = _randn(g, (b, 16, f, h, w)).to(dtype)
    ctx_posi = _randn(g, (1, TINY_CTX_LEN, TINY["text_dim"])).to(dtype)
    ctx_nega = _randn(g, (1, TINY_CTX_LEN, TINY["text_dim"])).to(dtype)
    return dict(latents=latents, y=y, control=control, ctx_posi=ctx_posi, ctx_nega=ctx_nega)


def block_inputs(dim, tokens, ctx_len, seed, dtype=torch.bfloat16):
    """Inputs of one DiT block as BASELINE config 1 prescribes: randn x/context, 0.5*randn t_mod."""
    g = torch.Generator().manual_seed(seed)
    x = _randn(g, (1, tokens, dim)).to(dtype)
    ctx = _randn(g, (1, ctx_len, dim)).to(dtype)
    t_mod = _randn(g, (1, 6, dim), 0.5).to(dtype)
    return x, ctx, t_mod


def vae_decoder_sd(names, shapes, seed, dtype=torch.bfloat16):
    """Seeded random weights for the whole VAE (encoder, conv1, conv2, decoder); `names`/`shapes` come from the
    fixture (they were read from the reference's own state_dict, in its order)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, shp in zip(names, shapes):
        shp = tuple(int(v) for v in shp if int(v) > 0)
        if n.endswith("gamma"):
            t = 1.0 + _randn(g, shp, 0.1)
        elif n.endswith("bias"):
            t = _randn(g, shp, 0.02)
        else:
            t = _randn(g, shp, 1.0 / math.sqrt(math.prod(shp[1:])))
        sd[n] = t.to(dtype)
    return sd


T5_TINY = dict(vocab=100, dim=256, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=2, num_buckets=32, shared_pos=False)


def t5_sd(cfg, seed, dtype=torch.bfloat16):
    """Seeded weights with the reference WanTextEncoder's key names (wan_video_text_encoder.py:209-243)."""
    g = torch.Generator().manual_seed(seed)
    d, da, df = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"]
    sd = {"token_embedding.weight": _randn(g, (cfg["vocab"], d)), "norm.weight": 1.0 + _randn(g, (d,), 0.1)}
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + _randn(g, (d,), 0.1)
        sd[p + "norm2.weight"] = 1.0 + _randn(g, (d,), 0.1)
        for n, shp, std in (("attn.q", (da, d), d ** -0.5), ("attn.k", (da, d), d ** -0.5), ("attn.v", (da, d), d ** -0.5),
                            ("attn.o", (d, da), da ** -0.5), ("ffn.gate.0", (df, d), d ** -0.5), ("ffn.fc1", (df, d), d ** -0.5),
                            ("ffn.fc2", (d, df), df ** -0.5)):
            sd[p + n + ".weight"] = _randn(g, shp, std)
        sd[p + "pos_embedding.embedding.weight"] = _randn(g, (cfg["num_buckets"], cfg["num_heads"]), 0.5)
    return {k: v.to(dtype) for k, v in sd.items()}


TRAIN_TIMESTEP_ID = 137     # the pinned "random" draw of training_loss (GF:183) in the training golden


def train_inputs(seed=77, dtype=torch.bfloat16):
    """Inputs of one training_loss call at the tiny size (GF:180-193): clean latents, noise, prompt embedding, image
    conditioning y (mask channels {0,1}), control latents."""
    g = torch.Generator().manual_seed(seed)
    b, c, f, h, w = TINY_LATENT
    y = _randn(g, (b, 20, f, h, w))
    y[:, :4] = (y[:, :4] > 0).float()
    return dict(input_latents=_randn(g, (b, c, f, h, w)).to(dtype), noise=_randn(g, (b, c, f, h, w)).to(dtype),
                context=_randn(g, (1, TINY_CTX_LEN, TINY["text_dim"])).to(dtype), y=y.to(dtype),
                control=_randn(g, (b, 16, f, h, w)).to(dtype))


def grad_probes(n: int, seed: int, k: int = 4) -> torch.Tensor:
    """k seeded unit-variance probe vectors [k, n] (fp64): projections of a gradient pin the whole tensor cheaply."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn((k, n), generator=g, dtype=torch.float64)


def grad_sample_index(n: int, seed: int, k: int = 256) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (k,), generator=g)


def checksum(tensors) -> float:
    """Order-dependent fp64 checksum of a dict/list of tensors (detects RNG / generation drift)."""
    items = tensors.items() if isinstance(tensors, dict) else enumerate(tensors)
    acc = 0.0
    for i, (k, t) in enumerate(items):
        acc += (i + 1) * float(t.double().sum()) + 0.5 * float(t.double().abs().sum())
    return acc


def same_checksum(a: float, b) -> bool:
    """fp64 sums are reduced in a machine-dependent order; compare to 1e-10 relative."""
    return math.isclose(float(a), float(b), rel_tol=1e-10)


def to_u16(t: torch.Tensor) -> np.ndarray:
    assert t.dtype == torch.bfloat16
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def from_u16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


PRELOOP_SHAPES = ((480, 832, 81), (470, 830, 80), (64, 96, 9), (17, 33, 6))


def preloop_inputs(seed=91):
    """Inputs of the pre-loop units (g10): a 50x70 RGB PIL image (so the unit's resize to 96x64 does work) and a
    [9,64,96,3] control video in [0,1] with two gaussian blobs (the value range the force maps have, DS:775-889)."""
    from PIL import Image
    g = torch.Generator().manual_seed(seed)
    img = (torch.rand((50, 70, 3), generator=g) * 255).to(torch.uint8).numpy()
    yy, xx = torch.meshgrid(torch.arange(64.0), torch.arange(96.0), indexing="ij")
    frames = []
    for f in range(9):
        a = torch.exp(-((xx - 20 - 4 * f) ** 2 + (yy - 30) ** 2) / (2 * 6.0 ** 2))
        b = torch.exp(-((xx - 70) ** 2 + (yy - 12 - 3 * f) ** 2) / (2 * 4.0 ** 2))
        frames.append(torch.stack([a, b, 0.5 * (a + b)], dim=-1))
    control = torch.stack(frames).clamp(0, 1).to(torch.bfloat16)
    return Image.fromarray(img), control


# g13: the keyword arguments of the whole-pipeline call (the reference's own names, GF:599-661) and its two prompts
PIPELINE_PROMPTS = ("the pendulum swings", "static, blurry")
PIPELINE_CALL_KWARGS = dict(seed=0, height=64, width=96, num_frames=9, num_inference_steps=3, cfg_scale=5.0, tiled=True,
                            tile_size=(6, 8), tile_stride=(3, 4), controlnet=True)


def preloop_decoded_video(seed=92):
    """A [1,3,2,8,8] 'decoded video' slightly outside [-1,1] (the clip must act) for the frames-to-uint8 conversion."""
    return (torch.rand((1, 3, 2, 8, 8), generator=torch.Generator().manual_seed(seed)) * 2.4 - 1.2).to(torch.bfloat16)


def vae_tile_inputs(seed=93):
    """One production-size VAE tile: latent [1,16,3,30,52] (normalised-latent scale) and a video [1,3,9,240,416] in [-1,1]
    (smooth low-frequency content + noise, so the encoder sees image-like statistics)."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn((1, 16, 3, 30, 52), generator=g).to(torch.bfloat16)
    yy, xx = torch.meshgrid(torch.linspace(0, 3.0, 240), torch.linspace(0, 5.0, 416), indexing="ij")
    frames = [torch.stack([torch.sin(xx + 0.3 * f + c) * torch.cos(yy - 0.2 * f) for c in range(3)]) for f in range(9)]
    vid = torch.stack(frames, dim=1)[None] * 0.8 + 0.1 * torch.randn((1, 3, 9, 240, 416), generator=g)
    return z, vid.clamp(-1, 1).to(torch.bfloat16)


FP8_CASES = ((72, 256, 256), (300, 528, 384), (515, 1024, 2048))


def fp8_case(M, N, K):
    """Seeded x [M,K], w [N,K], b [N] (bf16) for the fp8_linear contract (VRAM:115-151): activations of std 3 with one
    entry of 1500 (its row then has scale_a = 1500/448 > 1), one row of exact zeros and one row whose maximum is
    exactly 448 (the clamp's boundary)."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = (torch.randn((M, K), generator=g) * 3).to(torch.bfloat16)
    x[M // 2, 5] = 1500.0
    x[1] = 0
    x[2] = x[2].clamp(-400, 400)
    x[2, 7] = 448.0
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = (0.1 * torch.randn(N, generator=g)).to(torch.bfloat16)
    return x, w, b


def canny_inputs(seed=141):
    """g14: seeded uint8 clips for the Canny control video (DS:559-578): piecewise-constant random fields (16-pixel cells, lightly
    blurred: step edges of every height and both thresholds' sides) at the production frame size 480 x 832 and at 240 x 416 — both
    below the detector's 512 resolution, as every frame the dataset produces is (enlarged by Lanczos, edges reduced back by area)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = {}
    for tag, (h, w), n in (("240x416", (240, 416), 3), ("480x832", (480, 832), 2)):
        frames = []
        for _ in range(n):
            low = rng.random((h // 16 + 1, w // 16 + 1, 3))
            big = np.kron(low, np.ones((16, 16, 1)))[:h, :w]
            k = np.array([0.25, 0.5, 0.25])
            for ax in (0, 1):
                big = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, big)
            frames.append((big * 255).clip(0, 255).astype(np.uint8))
        out[tag] = np.stack(frames)
    return out


# g15: rows for the dominos / plants dataset variants (DS:1099-1894).  Dominos rows = g7's two synthetic rows (the reference's classes are
# line-for-line copies: the goldens must coincide); plants rows (force, angle, x_pos, y_pos, frames, height, width), the last at the
# signature's own defaults (49 frames, 480 x 720, DS:1667)
PLANTS_ROWS = [(200.0, 45.0, 0.25, 0.25, 81, 480, 832), (390.0, 200.0, 0.84, 0.83, 81, 480, 832), (35.0, -90.0, 0.5, 0.9, 81, 480, 832),
               (120.0, 10.0, 0.3, 0.6, 49, 480, 720)]


def training_clip(seed=161, frames=9, height=64, width=96):
    """g16: a 9-frame 96x64 RGB clip (list of PIL images) — a smooth random field drifting two pixels per frame."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    base = np.kron(rng.random((height // 8 + 4, width // 8 + 4, 3)), np.ones((8, 8, 1)))
    return [Image.fromarray((base[8:8 + height, 2 * i:2 * i + width] * 255).astype(np.uint8)) for i in range(frames)]


# g17: the training loop.  launch_training_task refuses anything but 832x480 frames (utils.py:653-679), so the items are full-size, 5 frames
TRAIN_LOOP = dict(num_frames=5, learning_rate=1e-3, weight_decay=1e-2, save_steps=3, num_epochs=2, max_grad_norm=1.0,
                  max_timestep_boundary=0.358, min_timestep_boundary=0.0, remove_prefix_in_ckpt="pipe.dit.",
                  control_signal_type="direct_force_and_goal_force_and_mass")


def training_items(n=4, frames=5, height=480, width=832, seed=171):
    """n training items as the datasets hand them over (DS:1488-1539): {"video": list of PIL frames, "prompt", "control_video"
    [F,H,W,3] in [0,1] bf16, "file_id"}: drifting smooth random fields and two moving gaussian blobs."""
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    yy, xx = torch.meshgrid(torch.arange(float(height)), torch.arange(float(width)), indexing="ij")
    items = []
    for k in range(n):
        base = np.kron(rng.random((height // 16 + 4, width // 16 + 4, 3)), np.ones((16, 16, 1)))
        video = [Image.fromarray((base[16:16 + height, 4 * i:4 * i + width] * 255).astype(np.uint8)) for i in range(frames)]
        cx, cy = 100.0 + 150 * k, 120.0 + 60 * k
        ctrl = []
        for f in range(frames):
            a = torch.exp(-((xx - cx - 12 * f) ** 2 + (yy - cy) ** 2) / (2 * 20.0 ** 2))
            b = torch.exp(-((xx - 600) ** 2 + (yy - 300 + 8 * f) ** 2) / (2 * 12.0 ** 2))
            ctrl.append(torch.stack([a, b, 0.5 * b], dim=-1))
        items.append({"video": video, "prompt": PIPELINE_PROMPTS[k % 2], "control_video": torch.stack(ctrl).clamp(0, 1).to(torch.bfloat16),
                      "file_id": f"item{k}"})
    return items


def train_loop_draws(step, latent_shape=(1, 16, 2, 60, 104), lo=0, hi=358):
    """The random draws of training step `step` as g17 pinned them: the generator made the reference's forward start from
    torch.manual_seed(1000 + step), after which NoiseInitializer draws the noise (fp32, CPU; GF:757-760) and training_loss the
    timestep id (GF:183)."""
    torch.manual_seed(1000 + step)
    noise = torch.randn(latent_shape, dtype=torch.float32)
    tid = int(torch.randint(lo, hi, (1,)))
    return noise, tid


def write_training_tree(root):
    """g18: a synthetic training-data tree as scripts/train/*.sh lay it out — three folders of clips with their CSV listings.  The clips
    are EMPTY files named *.mp4 (load_metadata only checks that they exist); one CSV row per set names a clip that is missing."""
    import os
    hdr = ("video,projectile_force_angle,projectile_force_magnitude,projectile_coordx,projectile_coordy,projectile_mass,"
           "target_indirect_force_angle,target_indirect_force_magnitude,target_coordx,target_coordy,target_mass,width,height,caption\n")
    sets = {"balls": [("b0.mp4", 10, 120.0, 100, 200, 1.5, 30, 80.5, 400, 210, 2.0), ("b1.mp4", 200, 390.5, 300, 100, 3.5, -1, -1, 500, 90, -1),
                      ("b2.mp4", 45, 33.25, 50, 60, 1.0, 350, 260.0, 70, 80, 4.0), ("b_missing.mp4", 1, 9999.0, 1, 1, 9.0, 1, 9999.0, 1, 1, 9.0)],
            "dominos": [("d0.mp4", 0, 55.0, 10, 20, 2.0, 0, 66.0, 30, 40, 2.5), ("d1.mp4", 90, 75.0, 11, 21, 2.25, 180, 44.0, 31, 41, 3.0),
                        ("d_missing.mp4", 1, 1.0, 1, 1, 0.5, 1, 1.0, 1, 1, 0.5)]}
    for name, rows in sets.items():
        os.makedirs(os.path.join(root, name), exist_ok=True)
        with open(os.path.join(root, name + ".csv"), "w") as f:
            f.write(hdr)
            for r in rows:
                f.write(f'{r[0]},{r[1]},{r[2]},{r[3]},{r[4]},{r[5]},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},832,480,"a caption"\n')
                if "missing" not in r[0]:
                    open(os.path.join(root, name, r[0]), "w").close()
    os.makedirs(os.path.join(root, "plants"), exist_ok=True)
    with open(os.path.join(root, "plants.csv"), "w") as f:
        f.write("video,force,angle,coordx,coordy,width,height,caption\n")
        for r in (("fern0.mp4", 12.5, 30, 100, 100), ("carnation1.mp4", 48.0, 200, 400, 240), ("p_missing.mp4", 999.0, 1, 1, 1)):
            f.write(f'{r[0]},{r[1]},{r[2]},{r[3]},{r[4]},832,480,"a plant"\n')
            if "missing" not in r[0]:
                open(os.path.join(root, "plants", r[0]), "w").close()


def training_cli(root):
    """g18: the command line of scripts/train/train_goal_force.sh on the tree of write_training_tree(root)."""
    import os
    j = lambda *a: os.path.join(root, *a)
    return ["--dataset_base_path", j("balls"), j("dominos"), j("plants"), "--dataset_metadata_path", j("balls.csv"), j("dominos.csv"), j("plants.csv"),
            "--control_signal_type", "direct_force_and_goal_force_and_mass", "--controlnet_num_layers", "10", "--height", "480", "--width", "832",
            "--num_frames", "81", "--dataset_repeat", "1", "--learning_rate", "1e-5", "--num_epochs", "2", "--save_steps", "500",
            "--remove_prefix_in_ckpt", "pipe.dit.", "--trainable_models", "controlnet", "--output_path", j("out"), "--extra_inputs", "input_image",
            "--max_timestep_boundary", "0.358", "--min_timestep_boundary", "0", "--max_grad_norm", "1", "--p_mask_out_masses", "0.5",
            "--p_mask_out_direct_force", "0.5", "--p_mask_out_indirect_force", "0.5", "--wandb_logging"]


def shape_check_cases():
    """g18: items for `data_is_correct_shape_and_type` (utils.py:653-679) — name, item, num_frames."""
    from PIL import Image
    ok = Image.new("RGB", (832, 480))
    small = Image.new("RGB", (416, 240))
    cv = lambda f, h=480, w=832, c=3: torch.zeros((f, h, w, c), dtype=torch.bfloat16)
    return [("good", {"video": [ok] * 5, "control_video": cv(5)}, 5), ("one small frame", {"video": [ok, small, ok], "control_video": cv(3)}, 3),
            ("control too short", {"video": [ok] * 5, "control_video": cv(4)}, 5), ("control wrong size", {"video": [ok] * 5, "control_video": cv(5, 240, 416)}, 5),
            ("frames are arrays", {"video": [torch.zeros(3, 480, 832)] * 5, "control_video": cv(5)}, 5),
            ("more frames than the control video checks", {"video": [ok] * 9, "control_video": cv(5)}, 5)]
