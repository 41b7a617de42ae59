"""model_fn_wan_video — one noise prediction of the Goal-Force sampler (the reference's own swap point
`pipe.model_fn`, src/goal_force/wan_video_new.py:161, 1349-1591), rebuilt on the HIP kernels.

Same keyword interface as the reference; branches the Goal-Force inference scripts never take
(S2V, VACE, camera/motion control, TeaCache, sliding window, reference latents, cfg-merged batches)
raise NotImplementedError instead of silently doing something else.

Differences that do not change results:
  * RoPE tables live on the device and are built once per grid (the reference rebuilds them on the CPU
    and copies 33.5 MB host->device every forward, GF:1474-1478);
  * ControlNet block i runs just before DiT block i (same arithmetic, but no list of 10 saved states);
  * zero-conv + residual add is one GEMM epilogue;
  * a ControlNet whose zero-convs are all exactly zero is skipped (x + 0 == x bitwise) unless
    `elide_zero_controlnet=False`;
  * `use_unified_sequence_parallel` / `sequence_parallel=` run the blocks on this rank's token chunk with head-parallel
    attention (sequence_parallel.py); unlike the reference's USP path the ControlNet tokens are sharded too;
  * `context_cache` (optional) memoises text_embedding(context) and the per-block cross-attention K/V,
    which are constant over the denoising steps for one (expert, prompt);
  * `cfg_shared` (optional, a dict owned by the caller for ONE denoising step): the cond and the uncond forward of a step differ
    only in the text context, which first enters a block at its cross-attention — the self-attention half of block 0 of the DiT
    and of the ControlNet sees identical inputs in both.  The first forward of the step stores those two tensors in the dict,
    the second takes them (DiTBlock.forward, `self_attn_memo`): bit-identical results, 2 of 100 self-attentions per step saved.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from ._lib import GoalForceError
from .dit import WanModel, pad_run


class ContextCache:
    """Per (expert, prompt) memo of text_embedding(context) and the cross-attention K/V of every block
    of the DiT and of its ControlNet (GF:1447, DIT:177-179 are recomputed each call in the reference)."""

    def __init__(self):
        self.ctx = None
        self.pad_n = None       # dit.pad_run(ctx): where the run of identical padded rows starts — detected once, used by every block
        self.dit_kv = {}
        self.cn_kv = {}


def _unsupported(name, value, default=None):
    if value is not None and value is not default and value is not False:
        raise NotImplementedError(f"model_fn_wan_video: `{name}` is outside the Goal-Force sampling path "
                                  "(SURVEY.md §2 #2)")


@torch.no_grad()
def model_fn_wan_video(
    dit: WanModel,
    motion_controller=None,
    vace=None,
    latents: torch.Tensor = None,
    timestep: torch.Tensor = None,
    context: torch.Tensor = None,
    clip_feature: Optional[torch.Tensor] = None,
    y: Optional[torch.Tensor] = None,
    reference_latents=None,
    vace_context=None,
    vace_scale=1.0,
    audio_embeds: Optional[torch.Tensor] = None,
    motion_latents: Optional[torch.Tensor] = None,
    s2v_pose_latents: Optional[torch.Tensor] = None,
    drop_motion_frames: bool = True,
    tea_cache=None,
    use_unified_sequence_parallel: bool = False,
    motion_bucket_id: Optional[torch.Tensor] = None,
    sliding_window_size: Optional[int] = None,
    sliding_window_stride: Optional[int] = None,
    cfg_merge: bool = False,
    use_gradient_checkpointing: bool = False,
    use_gradient_checkpointing_offload: bool = False,
    control_camera_latents_input=None,
    fuse_vae_embedding_in_latents: bool = False,
    controlnet=None,
    context_cache: Optional[ContextCache] = None,
    elide_zero_controlnet: bool = True,
    sequence_parallel=None,
    cfg_shared: Optional[dict] = None,
    **kwargs,
):
    for name, val in (("motion_controller", motion_controller), ("vace", vace), ("reference_latents", reference_latents),
                      ("vace_context", vace_context), ("audio_embeds", audio_embeds), ("tea_cache", tea_cache),
                      ("motion_bucket_id", motion_bucket_id), ("sliding_window_size", sliding_window_size),
                      ("control_camera_latents_input", control_camera_latents_input),
                      ("clip_feature", clip_feature if dit.require_clip_embedding else None)):
        _unsupported(name, val)
    sp = sequence_parallel
    if use_unified_sequence_parallel and sp is None:
        from .sequence_parallel import SequenceParallel
        sp = SequenceParallel()          # the reference's flag: Ulysses over the whole world group (GF:1422-1426)
    if sp is not None and sp.size == 1:
        sp = None
    if cfg_merge or latents.shape[0] != 1 or context.shape[0] != 1:
        raise NotImplementedError("cfg_merge / batch > 1: run the cond and uncond forwards separately (GF:710-716)")
    if dit.seperated_timestep and fuse_vae_embedding_in_latents:
        raise NotImplementedError("seperated_timestep is a Wan2.2-TI2V-5B feature")
    if not latents.is_cuda:
        raise GoalForceError("model_fn_wan_video: tensors must be on the GPU (no CPU fallback exists)")

    use_controlnet = controlnet is not None

    # Timestep (GF:1441-1442)
    t, t_mod = dit.time_embed(timestep)

    # Text embedding (GF:1447)
    if context_cache is not None and context_cache.ctx is not None:
        ctx, pad_n = context_cache.ctx, context_cache.pad_n
    else:
        ctx = dit.embed_text(context)
        # ONE read-back per embedded context; the blocks below take n instead of detecting the run again
        pad_n = pad_run(ctx[0]) if ops._OPT["fold_pad_keys"] else None
        if context_cache is not None:
            context_cache.ctx, context_cache.pad_n = ctx, pad_n

    # Image embedding + patchify (GF:1456-1464) — cat([latents, y]) is folded into the patch gather
    x, (f, h, w) = dit.patchify(latents, extra=y if (y is not None and dit.require_vae_embedding) else None)
    x = x[0]
    rope = dit.rope_table(f, h, w, latents.device)
    if sp is not None:                      # GF:1526-1531: contiguous token chunks; RoPE of the chunk (xdit:36-37)
        x = sp.shard_tokens(x).contiguous()
        rope = dit.rope_table_shard(f, h, w, latents.device, sp)

    run_cn = use_controlnet and not (elide_zero_controlnet and controlnet.all_zero())
    c = None
    if run_cn:
        if controlnet.stride is not None:
            raise NotImplementedError("strided ControlNet (apply_strided_controlnet) is not used by Goal Force")
        control_latents = kwargs.get("control_signal_video_latents", None)
        if control_latents is None:
            raise GoalForceError("controlnet given but control_signal_video_latents missing")
        c = controlnet.controlnet_patch_embedding(control_latents)[0]      # GF:1493
        if sp is not None:
            c = sp.shard_tokens(c).contiguous()   # ControlNet tokens follow the DiT's chunking (injection stays local)
        n_cn = controlnet.controlnet_dit.num_layers
    else:
        n_cn = 0

    def kv_for(cache_dict, block, idx):
        if context_cache is None:
            return None
        if idx not in cache_dict:
            cache_dict[idx] = block.cross_attn.context_kv(ctx[0], pad_n=pad_n)
        return cache_dict[idx]

    # blocks (GF:1503-1570); ControlNet block i is evaluated right before DiT block i
    for block_id, block in enumerate(dit.blocks):
        memo_cn = memo_dit = None
        if cfg_shared is not None and block_id == 0:     # see the module docstring: identical inputs in both CFG branches
            memo_cn, memo_dit = cfg_shared.setdefault("cn0", {}), cfg_shared.setdefault("dit0", {})
        if block_id < n_cn:
            cb = controlnet.controlnet_dit.blocks[block_id]
            c = cb(c, ctx, t_mod, rope, context_kv=kv_for(context_cache.cn_kv if context_cache else None, cb, block_id),
                   sp=sp, self_attn_memo=memo_cn, pad_n=pad_n)
        x = block(x, ctx, t_mod, rope, context_kv=kv_for(context_cache.dit_kv if context_cache else None, block, block_id),
                  sp=sp, self_attn_memo=memo_dit, pad_n=pad_n)
        if block_id < n_cn:
            # x = x + zero_conv(state)   (GF:1565-1570) — Conv1d(k=1) == Linear, fused residual epilogue
            ops.gemm(c, controlnet.zero_conv_weight(block_id), controlnet.controlnet_zero_convs_after[block_id].bias,
                     epilogue=ops.EPI_BIAS_RESID, resid=x, out=x)

    x = dit.head(x.unsqueeze(0), t)          # GF:1581
    if sp is not None:
        x = sp.gather_tokens(x[0], total=f * h * w).unsqueeze(0)   # GF:1582-1585; pad rows of a ragged split cut off (xdit:103)
    return dit.unpatchify(x, (f, h, w))      # GF:1590
