"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  `WanVideoPipeline.__call__` (src/goal_force/wan_video_new.py "GF":598-737) as one
function over the other oracle modules: the order in which the reference's 17 units and its loop touch the data, for the inputs
Goal Force gives it (prompt, negative prompt, input image, control-signal video, controlnet=True).

  set_timesteps                                GF:663           -> wan_oracle.flow_match_sigmas (inside denoise_loop)
  ShapeChecker                                 GF:741-747       -> preloop_oracle.shape_check
  NoiseInitializer                             GF:751-763       -> preloop_oracle.noise
  PromptEmbedder                               GF:808-820       -> `encode_prompt(prompt)`: a callable of the caller (the text encoder has
                                                                   its own oracle, t5_oracle.py); under CFG the negative prompt second
  InputVideoEmbedder (input_video is None)     GF:774-775       -> latents = noise
  ControlVideoEmbedder                         GF:791-805       -> preloop_oracle.control_latents
  ImageEmbedderVAE                             GF:887-917       -> preloop_oracle.image_y
  every other unit returns {} for these inputs (S2V GF:1206, CLIP / Fused / FunControl / FunReference / FunCameraControl /
  SpeedControl / VACE / UnifiedSequenceParallel / TeaCache / CfgMerger: their inputs are None / False)
  denoising loop, expert switch, CFG, step     GF:697-723       -> wan_oracle.denoise_loop
  tiled decode                                 GF:733           -> vae_oracle.tiled_decode (tile sizes in latent units, VAE:1245-1246)
  frames                                       GF:735, UTIL:76-91 -> frames_uint8 below

Pinned against tests/golden/g13_pipeline_call.npz — the outputs of the reference's OWN `__call__`, executed in the build container
by tests/golden/make_goldens.py::g13_pipeline_call — by tests/test_oracle_goldens.py::test_oracle_pipeline_call_vs_reference_call
(fp32 mode: latents to 1e-4, frames within one level; bf16 mode: the bars of the other bf16 goldens)."""
import torch

from . import preloop_oracle as po
from . import vae_oracle as vo
from . import wan_oracle as wo


def frames_uint8(video, min_value=-1, max_value=1):
    """BasePipeline.vae_output_to_video (UTIL:85-91) without the PIL wrapper: [1,3,T,H,W] -> uint8 [T,H,W,3]; the scale, the clip
    and the truncating cast act in the video's own dtype (UTIL:78-82)."""
    v = video.mean(dim=0).permute(1, 2, 3, 0)                                   # 'B C T H W -> T H W C' with reduce-mean over B
    return ((v - min_value) * (255 / (max_value - min_value))).clip(0, 255).to(torch.uint8)


def pipeline_call(experts, vae_sd, encode_prompt, prompt, negative_prompt, input_image, control_signal_video, seed, height, width,
                  num_frames, num_inference_steps=50, cfg_scale=5.0, tiled=True, tile_size=(30, 52), tile_stride=(15, 26),
                  sigma_shift=5.0, switch_DiT_boundary=0.875, dtype=torch.bfloat16):
    """-> (final latents [1,16,f,H/8,W/8], decoded video [1,3,F,H,W], uint8 frames [F,H,W,3]).
    experts = [(dit_sd, cfg, controlnet_sd, n_layers), (dit2_sd, cfg, controlnet2_sd, n_layers)] as wan_oracle.denoise_loop takes
    them; vae_sd in `dtype`; encode_prompt(prompt) -> [1,L,text_dim]."""
    height, width, num_frames = po.shape_check(height, width, num_frames)
    noise = po.noise(height, width, num_frames, seed, dtype=dtype)
    ctx_posi = encode_prompt(prompt).to(dtype)
    ctx_nega = encode_prompt(negative_prompt).to(dtype) if cfg_scale != 1.0 else None       # UTIL:262-271
    control = po.control_latents(control_signal_video.to(dtype), vae_sd, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)
    y = po.image_y(input_image, num_frames, height, width, vae_sd, dtype=dtype, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)
    latents = wo.denoise_loop(experts, noise, ctx_posi, ctx_nega, y, control.to(dtype), num_inference_steps, cfg_scale=cfg_scale,
                              shift=sigma_shift, boundary=switch_DiT_boundary, dtype=dtype)
    video = vo.tiled_decode(latents, vae_sd, tile_size, tile_stride) if tiled else vo.decode(latents, vae_sd)
    return latents, video, frames_uint8(video)
