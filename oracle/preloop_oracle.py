"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Restatement of the three pre-loop conditioning units of the Goal-Force
pipeline (src/goal_force/wan_video_new.py "GF") on torch-CPU, over oracle/vae_oracle.py's encoder:

  * noise            WanVideoUnit_NoiseInitializer GF:751-763 -> BasePipeline.generate_noise UTIL:117-122
  * control_latents  WanVideoUnit_ControlVideoEmbedder GF:791-805
  * image_y          WanVideoUnit_ImageEmbedderVAE GF:887-917 (the `end_image is None` branch; preprocess_image UTIL:60-66)
  * shape_check      WanVideoUnit_ShapeChecker GF:740-747 -> check_resize_height_width UTIL:41-57
  * input_latents    WanVideoUnit_InputVideoEmbedder GF:767-789 up to its VAE encode (preprocess_video UTIL:69-73): the training
                     branch's `input_latents`; pinned bit-exactly (bf16) against tests/golden/g16_forward_preprocess.npz = the
                     reference's own units run in training mode as scripts/train/train.py:76-118 runs them

Pinned bit-exactly (bf16) against tests/golden/g10_preloop.npz, which tests/golden/make_goldens.py::g10_preloop made by
running the REFERENCE's unit classes on the reference's BasePipeline and WanVideoVAE (tests/test_preloop.py).
Tile sizes are given in latent units like the reference's `tile_size` / `tile_stride` arguments."""
import numpy as np
import torch

from . import vae_oracle as vo


def shape_check(height, width, num_frames, hdiv=16, wdiv=16, tdiv=4, trem=1):
    """Round height/width up to multiples of 16 and num_frames up to 4k+1 (UTIL:41-57 with the factors of GF:123-126)."""
    height += -height % hdiv
    width += -width % wdiv
    if num_frames % tdiv != trem:
        num_frames = -(-num_frames // tdiv) * tdiv + trem
    return height, width, num_frames


def noise(height, width, num_frames, seed, dtype=torch.bfloat16, z_dim=16, up=8):
    """fp32 normal draws from a CPU generator seeded with `seed`, rounded to `dtype` (GF:757-760, UTIL:117-122)."""
    shape = (1, z_dim, (num_frames - 1) // 4 + 1, height // up, width // up)
    return torch.randn(shape, generator=torch.Generator("cpu").manual_seed(seed), dtype=torch.float32).to(dtype)


def _encode(video, sd, tiled, tile_size, tile_stride):
    if not tiled:
        return vo.encode(video, sd)
    px = lambda t: (t[0] * 8, t[1] * 8)                       # VAE:1245-1246: latent units -> pixels
    return vo.tiled_encode(video, sd, px(tile_size), px(tile_stride))


def control_latents(control_signal_video, sd, tiled=True, tile_size=(30, 52), tile_stride=(15, 26)):
    """[F,H,W,3] force-map video in [0,1], NOT rescaled to [-1,1] -> latents [1,16,f,H/8,W/8] (GF:799-803)."""
    video = control_signal_video.permute(3, 0, 1, 2).unsqueeze(0)
    return _encode(video, sd, tiled, tile_size, tile_stride)


def image_y(image, num_frames, height, width, sd, dtype=torch.bfloat16, tiled=True, tile_size=(30, 52),
            tile_stride=(15, 26)):
    """PIL image -> y [1,20,f,H/8,W/8]: 4 mask channels (ones on latent frame 0, zeros after: the first pixel frame is
    repeated 4x and the 4k+... pixel frames are regrouped into 4 channels per latent frame, GF:899-909) followed by the
    16 latent channels of vae.encode([image, 0, 0, ...]) (GF:906, 911-913)."""
    px = torch.from_numpy(np.array(image.resize((width, height)), dtype=np.float32)).to(dtype)
    px = px * (2 / 255) + (-1)                                                  # UTIL:63-64, arithmetic in `dtype`
    video = torch.zeros((1, 3, num_frames, height, width), dtype=dtype)
    video[0, :, 0] = px.permute(2, 0, 1)
    f = (num_frames - 1) // 4 + 1
    mask = torch.zeros((4, f, height // 8, width // 8), dtype=dtype)
    mask[:, 0] = 1
    lat = _encode(video, sd, tiled, tile_size, tile_stride)[0]
    return torch.cat([mask, lat.to(dtype)]).unsqueeze(0)


def input_latents(frames, sd, dtype=torch.bfloat16, tiled=False, tile_size=(30, 52), tile_stride=(15, 26)):
    """list of PIL frames -> latents [1,16,f,H/8,W/8] of the clip (GF:774-776): every frame `x * (2 / 255) - 1` in `dtype`
    (UTIL:60-66), stacked along T (UTIL:69-73), through vae.encode."""
    px = [torch.from_numpy(np.array(f, dtype=np.float32)).to(dtype) * (2 / 255) + (-1) for f in frames]
    video = torch.stack([p.permute(2, 0, 1) for p in px], dim=1).unsqueeze(0)                 # [1,3,T,H,W]
    return _encode(video, sd, tiled, tile_size, tile_stride).to(dtype)
