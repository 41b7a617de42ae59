"""Full-size VAE parity tool (not collected by pytest): the tiled decode of a whole 832 x 480 x 81-frame video's latents and the
tiled encode of such a video — what WanVideoPipeline.__call__ runs after / before the loop (GF:733, 713-716; VAE:1103-1203) —
on the HIP path against the REFERENCE'S ARITHMETIC at the same size.

The reference's arithmetic here = oracle/vae_oracle.py (the torch restatement of diffsynth/models/wan_video_vae.py, pinned
bit-exactly to the reference's own WanVideoVAE by g6 / g12) run on this GPU through torch-ROCm's kernels (MIOpen convolutions,
SDPA), once in bf16 — what the reference would compute here — and once in fp32 — the yardstick both are measured against.
The goldens pin ONE production tile with 3 latent frames (g12); this tool runs all 9 tiles x 21 latent frames and the blend.

    python tests/fullsize_vae_parity.py                 # decode + encode, random-init VAE (seed 61 = the goldens' weights)
    python tests/fullsize_vae_parity.py --peaky 6       # the AttentionBlocks' q projection x 6: peaky frame attention

Output: rel-L2 of hip-bf16 / ref-bf16 against fp32 and against each other, PSNR of the uint8 frames (UTIL:76-91), the same for
the one-GEMM score form of rounds 1-5 (ops.options(vae_attn_offset=False)), and wall times (a yardstick: torch-ROCm's kernels
on the reference's per-latent-frame streaming schedule)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import gen_inputs as gi                      # noqa: E402
from oracle import vae_oracle as vo          # noqa: E402

BF = torch.bfloat16
TILE, STRIDE = (30, 52), (15, 26)            # latent units (GF:733)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def frames_u8(x):
    """UTIL:76-91 in the tensor's own dtype."""
    return ((x[0].permute(1, 2, 3, 0) + 1) * (255 / 2)).clip(0, 255).to(torch.uint8)


def psnr_u8(a, b):
    mse = float((a.double() - b.double()).pow(2).mean())
    return 99.0 if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)


def weights(peaky):
    g6 = np.load(os.path.join(ROOT, "tests", "golden", "g6_vae.npz"))
    sd = gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61)
    if peaky != 1.0:
        for side in ("decoder", "encoder"):
            for leaf in ("weight", "bias"):
                t = sd[f"{side}.middle.1.to_qkv.{leaf}"]
                C = t.shape[0] // 3
                t[:C] = (t[:C].float() * peaky).to(BF)                   # q rows: logits x peaky, rounded ONCE for every path
    return sd


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.time()
    out = fn()
    torch.cuda.synchronize()
    return out, time.time() - t0


def run(grid=(21, 60, 104), peaky=1.0, encode=True, log=print):
    from goal_force_amd import ops
    from goal_force_amd.vae import WanVideoVAE
    torch.set_grad_enabled(False)
    sd = weights(peaky)
    sd_bf = {k: v.cuda() for k, v in sd.items()}
    sd_32 = {k: v.float().cuda() for k, v in sd.items()}
    vae = WanVideoVAE()
    vae.load_state_dict({"model." + k: t for k, t in sd.items()}, strict=True)
    vae = vae.to(BF).cuda()
    T, H, W = grid
    z = torch.randn((1, 16, T, H, W), generator=torch.Generator().manual_seed(4242)).to(BF).cuda()
    px = lambda t: (t[0] * 8, t[1] * 8)
    rep = {"grid": list(grid), "peaky": peaky, "tile": TILE, "stride": STRIDE,
           "weights": "random init, seed 61 (the goldens' VAE weights)" + (f", AttentionBlock q projection x {peaky:g}" if peaky != 1.0 else "")}

    hip_dec = lambda: vae.decode(z, tiled=True, tile_size=TILE, tile_stride=STRIDE)
    hip_dec()                                                            # warm-up (pools, first launches)
    hip, t_hip = timed(hip_dec)
    with ops.options(vae_attn_offset=False):
        hip1, _ = timed(hip_dec)
    log(f"decode: hip {t_hip:.3f} s; running the reference arithmetic on torch-ROCm (bf16, then fp32) ...")
    vo.tiled_decode(z[:, :, :2], sd_bf, TILE, STRIDE)                    # warm-up of the library's kernel selection
    ref, t_ref = timed(lambda: vo.tiled_decode(z, sd_bf, TILE, STRIDE))
    f32, t_f32 = timed(lambda: vo.tiled_decode(z.float(), sd_32, TILE, STRIDE))
    u8 = {k: frames_u8(v) for k, v in (("hip", hip), ("hip1", hip1), ("ref", ref), ("f32", f32))}
    rep["decode"] = {
        "shape": list(hip.shape),
        "rel_l2": {"hip_bf16_vs_fp32": rel(hip, f32), "ref_bf16_vs_fp32": rel(ref, f32), "hip_bf16_vs_ref_bf16": rel(hip, ref),
                   "hip_one_gemm_scores_vs_fp32": rel(hip1, f32)},
        "psnr_db_uint8_frames": {"hip_bf16_vs_fp32": psnr_u8(u8["hip"], u8["f32"]), "ref_bf16_vs_fp32": psnr_u8(u8["ref"], u8["f32"]),
                                 "hip_bf16_vs_ref_bf16": psnr_u8(u8["hip"], u8["ref"]),
                                 "hip_one_gemm_scores_vs_fp32": psnr_u8(u8["hip1"], u8["f32"])},
        "seconds": {"hip": t_hip, "reference_arithmetic_on_torch_rocm_bf16": t_ref, "fp32": t_f32},
    }
    log(f"  {json.dumps(rep['decode'])}")
    if encode:
        video = f32.clamp(-1, 1).to(BF)                                  # a video with the decoder's own statistics
        hip_enc = lambda: vae.encode([video[0]], tiled=True, tile_size=TILE, tile_stride=STRIDE)
        hip_enc()
        hip, t_hip = timed(hip_enc)
        with ops.options(vae_attn_offset=False):
            hip1, _ = timed(hip_enc)
        vo.tiled_encode(video[:, :, :5], sd_bf, px(TILE), px(STRIDE))
        ref, t_ref = timed(lambda: vo.tiled_encode(video, sd_bf, px(TILE), px(STRIDE)))
        f32e, t_f32 = timed(lambda: vo.tiled_encode(video.float(), sd_32, px(TILE), px(STRIDE)))
        rep["encode"] = {
            "shape": list(hip.shape),
            "rel_l2": {"hip_bf16_vs_fp32": rel(hip, f32e), "ref_bf16_vs_fp32": rel(ref, f32e), "hip_bf16_vs_ref_bf16": rel(hip, ref),
                       "hip_one_gemm_scores_vs_fp32": rel(hip1, f32e)},
            "seconds": {"hip": t_hip, "reference_arithmetic_on_torch_rocm_bf16": t_ref, "fp32": t_f32},
        }
        log(f"  {json.dumps(rep['encode'])}")
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=3, default=[21, 60, 104], help="latent f, H/8, W/8 (default 832x480x81f)")
    ap.add_argument("--peaky", type=float, default=1.0, help="multiply the AttentionBlocks' q projection by this factor")
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rep = run(tuple(a.grid), a.peaky, not a.no_encode)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(rep, f, indent=1)
