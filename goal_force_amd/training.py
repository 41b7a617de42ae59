"""ControlNet training step on the HIP kernels.

Reference: `WanVideoPipeline.training_loss` (src/goal_force/wan_video_new.py:180-193) — sample a timestep, noise the
input latents, run `model_fn` with the ControlNet, MSE against `noise - latents`, weight by the scheduler's training
weight — driven by `launch_training_task` (src/goal_force/utils.py:734-826): AdamW over the ControlNet parameters
(`pipe.controlnet.*`: patch embedding, 10 DiT blocks, 10 zero-convs; GF:97-117), bf16 parameters, optional gradient
clipping, checkpoints `{"pipe.controlnet.<name>": tensor}`.

Design (MI355X-first, not a port of torch autograd):
  * torch.autograd is only the tape between a few coarse Functions; every FLOP of forward and backward runs in this
    repo's HIP kernels (GEMM, flash-attention fwd/bwd, row-op backward kernels, loss, AdamW).
  * one Function per DiT / ControlNet BLOCK: its forward is the fused inference path (GEMM epilogues etc.) and keeps only
    the block input; its backward re-runs the block un-fused to rebuild the intermediates, then walks them backwards
    (block-granular activation checkpointing is built in — what the reference reaches for with
    use_gradient_checkpointing, GF:1503-1557; at S=32760 a block's intermediates are ~6 GB, its input 335 MB).
  * weight gradients are only formed for parameters that require them (the DiT experts are frozen: their blocks cost
    a forward + an activation-only backward); GEMM backward reuses the forward GEMM on transposed operands
    (dX = dY W via W^T; dW = dY^T X via dY^T, X^T padded to the K step).
  * gradient fan-in (a tensor used twice) is summed by the HIP add kernel inside the Functions, not by autograd.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import GoalForceError
from .dit import Q_PRESCALE, DiTBlock, RopeTable, WanModel, _tokens2d

BF = torch.bfloat16


def _pad64(n: int) -> int:
    return -(-n // 64) * 64


def _zeros_f32(dim, device):
    return torch.zeros((dim,), dtype=torch.float32, device=device)


# ---------------------------------------------------------------------------------------------------------------------
# GEMM backward on the forward kernel
def linear_dx(dy: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """dX[M,K] = dY[M,N] W[N,K]  (W^T is the GEMM's [N',K'] operand; N must be a multiple of the 64-wide K step)."""
    n, k = weight.shape
    if n % 64:
        raise GoalForceError(f"linear_dx: out_features={n} must be a multiple of 64")
    w_t = ops.transpose_pad(weight.detach(), n)              # [K, N]
    return ops.gemm(dy, w_t, None)


def linear_dw(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dW[N,K] = dY[M,N]^T X[M,K]  (both operands transposed with the token dim zero-padded to a multiple of 64)."""
    m = dy.shape[0]
    mp = _pad64(m)
    return ops.gemm(ops.transpose_pad(dy, mp), ops.transpose_pad(x, mp), None)


def bias_grad(dy: torch.Tensor) -> torch.Tensor:
    acc = _zeros_f32(dy.shape[1], dy.device)
    ops.colsum(dy, acc=acc)
    return ops.f32_to_bf16(acc)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMM, with its backward."""

    @staticmethod
    def forward(ctx, x2, weight, bias):
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return ops.gemm(x2, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = linear_dx(dy, weight) if ctx.needs_input_grad[0] else None
        dw = linear_dw(dy, x2) if ctx.needs_input_grad[1] else None
        db = bias_grad(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


class PatchEmbedFn(torch.autograd.Function):
    """ControlNet_PatchEmbedding (GF:72-94): Conv3d k=s=(1,2,2) == GEMM over the gathered patches; `cols` [S, kpad]."""

    @staticmethod
    def forward(ctx, cols, conv_weight, bias):
        k = conv_weight.shape[1] * 4
        wp = torch.zeros((conv_weight.shape[0], cols.shape[1]), dtype=conv_weight.dtype, device=conv_weight.device)
        wp[:, :k] = conv_weight.detach().reshape(conv_weight.shape[0], k)
        ctx.save_for_backward(cols)
        ctx.wshape, ctx.k = conv_weight.shape, k
        return ops.gemm(cols, wp, bias)

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        dy = dy.contiguous()
        dw = linear_dw(dy, cols)[:, :ctx.k].reshape(ctx.wshape).contiguous() if ctx.needs_input_grad[1] else None
        db = bias_grad(dy) if ctx.needs_input_grad[2] else None
        return None, dw, db


class LayerNormModFn(torch.autograd.Function):
    """LayerNorm (no affine) * scale1p + shift (Head.forward, DIT:253-269); no parameter gradients needed there."""

    @staticmethod
    def forward(ctx, x2, scale1p, shift, eps):
        ctx.save_for_backward(x2, scale1p)
        ctx.eps = eps
        return ops.layernorm_modulate(x2, scale1p=scale1p, shift=shift, eps=eps)

    @staticmethod
    def backward(ctx, dy):
        x2, scale1p = ctx.saved_tensors
        return ops.layernorm_bwd(x2, dy.contiguous(), g=scale1p, eps=ctx.eps), None, None, None


class InjectFn(torch.autograd.Function):
    """x + zero_conv(state)  (GF:1565-1570; Conv1d(k=1) == Linear): one GEMM with the residual epilogue."""

    @staticmethod
    def forward(ctx, x2, state, weight2d, bias):
        ctx.save_for_backward(state, weight2d)
        return ops.gemm(state, weight2d, bias, epilogue=ops.EPI_BIAS_RESID, resid=x2)

    @staticmethod
    def backward(ctx, dout):
        state, w = ctx.saved_tensors
        dout = dout.contiguous()
        dstate = linear_dx(dout, w) if ctx.needs_input_grad[1] else None
        dw = linear_dw(dout, state) if ctx.needs_input_grad[2] else None
        db = bias_grad(dout) if ctx.needs_input_grad[3] else None
        return dout, dstate, dw, db


class FanoutFn(torch.autograd.Function):
    """A tensor consumed twice: the two gradients are summed by the HIP add kernel."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        return ops.add(g1.contiguous(), g2.contiguous())


class UnpatchifyFn(torch.autograd.Function):
    """'b (f h w) (x y z c) -> b c (f x) (h y) (w z)' (DIT:351-356); backward is the inverse data movement."""

    @staticmethod
    def forward(ctx, tokens, c, f, h, w):
        ctx.dims = (c, f, h, w)
        return ops.unpatchify(tokens.contiguous(), c, f, h, w)

    @staticmethod
    def backward(ctx, dout):
        c, f, h, w = ctx.dims
        g = dout.reshape(c, f, h, 2, w, 2).permute(1, 2, 4, 3, 5, 0)      # f h w y z c
        return g.reshape(f * h * w, 4 * c).contiguous(), None, None, None, None


class MseLossFn(torch.autograd.Function):
    """weight * F.mse_loss(pred.float(), target.float())  (GF:190-191)."""

    @staticmethod
    def forward(ctx, pred, target, weight):
        loss, dpred = ops.mse_loss(pred.contiguous(), target.contiguous(), weight=weight, want_grad=True)
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        if float(g) != 1.0:
            dpred = (dpred.float() * float(g)).to(BF)
        return dpred, None, None


# ---------------------------------------------------------------------------------------------------------------------
# one DiT / ControlNet block
_PARAM_NAMES = (
    "modulation",
    "self_attn.q.weight", "self_attn.q.bias", "self_attn.k.weight", "self_attn.k.bias", "self_attn.v.weight",
    "self_attn.v.bias", "self_attn.o.weight", "self_attn.o.bias", "self_attn.norm_q.weight", "self_attn.norm_k.weight",
    "norm3.weight", "norm3.bias",
    "cross_attn.q.weight", "cross_attn.q.bias", "cross_attn.k.weight", "cross_attn.k.bias", "cross_attn.v.weight",
    "cross_attn.v.bias", "cross_attn.o.weight", "cross_attn.o.bias", "cross_attn.norm_q.weight", "cross_attn.norm_k.weight",
    "ffn.0.weight", "ffn.0.bias", "ffn.2.weight", "ffn.2.bias",
)


# What a block's forward keeps for its backward besides its inputs (set_keep_level):
#   "none": nothing — the backward recomputes the whole block;
#   "attn": the self-attention output + log-sum-exp (0.34 GB per block at 32760 tokens; 17 GB at A14B size, 0.9 s less per step);
#   "wide": also the three self-attention projections (pre-norm q, k; v) and the block's state after the self- and the
#           cross-attention branch: 2.0 GB per block, 100 GB at A14B size — a training step then peaks near 190 GB of the 288 GB and
#           its backward skips 4.6 (trainable block) to 9.5 ms (frozen block: also o, GELU and FFN2, which only feed parameter
#           gradients) of recomputed GEMMs per block.  Kept tensors are the forward's own values: same kernels, same inputs.
#   "auto" (default): "wide" for a block whose forward finds the device with room for it (free memory, torch's cached blocks
#           included, >= WIDE_HEADROOM x the five tensors it would keep), else "attn" — a process that also holds the second
#           expert, a longer sequence or a smaller part degrades to the 17 GB setting instead of running out of memory.
KEEP_ATTENTION, KEEP_WIDE, KEEP_AUTO = True, False, True


def set_keep_level(level: str = "auto") -> str:
    """Choose what the blocks' forwards keep ("none" / "attn" / "wide" / "auto", above); returns the level that was in force.
    A call, not an environment variable: nothing in this package reads the environment to pick a code path."""
    global KEEP_ATTENTION, KEEP_WIDE, KEEP_AUTO
    if level not in ("none", "attn", "wide", "auto"):
        raise GoalForceError(f"set_keep_level({level!r}): expected none, attn, wide or auto")
    old = "none" if not KEEP_ATTENTION else "wide" if KEEP_WIDE else "auto" if KEEP_AUTO else "attn"
    KEEP_ATTENTION, KEEP_WIDE, KEEP_AUTO = level != "none", level == "wide", level == "auto"
    return old


WIDE_HEADROOM = 24      # blocks' worth of wide tensors that must still fit (the blocks ahead of this one + the backward's ~6 GB working set)


def _wide_fits(x2):
    """auto mode: is there room to keep this block's five wide tensors (5 x S x D bf16) — and WIDE_HEADROOM more like it?"""
    if KEEP_WIDE:
        return True
    if not KEEP_AUTO or not x2.is_cuda:
        return False
    free, _ = torch.cuda.mem_get_info(x2.device)
    free += torch.cuda.memory_reserved(x2.device) - torch.cuda.memory_allocated(x2.device)     # cached blocks torch can reuse
    return free >= WIDE_HEADROOM * 5 * x2.numel() * x2.element_size()


_WIDE_NAMES = ("qp", "kp", "v", "x1", "x2b")


def _block_params(block: DiTBlock):
    named = dict(block.named_parameters())
    return [named[n] for n in _PARAM_NAMES]


class DiTBlockFn(torch.autograd.Function):
    """DiTBlock.forward (DIT:197-230) with a hand-written backward; inputs x2 [S,D], ctx2 [L,D], t_mod [1,6,D]."""

    @staticmethod
    def forward(ctx, block, rope, x2, ctx2, t_mod, *params):
        ctx.block, ctx.rope = block, rope
        if any(getattr(m, "_gf_w8", None) is not None for m in block.modules()):
            # the backward recomputes the block on the bf16 kernels: an fp8 forward would not be the function differentiated
            raise GoalForceError("training through an enable_fp8 block is refused: call enable_fp8(module, False) first")
        ctx.param_needs = [p.requires_grad for p in params]
        ctx.q_prescale = bool(ops._OPT["attn_q_prescale"])       # what the forward's self-attention saw is what the backward rebuilds (dit.SelfAttention.attend)
        keep = ({"wide": True} if _wide_fits(x2) else {}) if KEEP_ATTENTION else None
        with torch.no_grad():
            out = block(x2, ctx2, t_mod, rope, keep=keep, fold_pad_keys=False)   # the backward differentiates the unfolded graph
        # kept for the backward besides the block's inputs: the self-attention output (S x D bf16) and its log-sum-exp
        # (S x heads fp32) — 0.34 GB per block at 32760 tokens against 18.6 ms of attention per block not run again.
        # What the forward actually stored decides the level (a forward that took a memo / sharded path stores less):
        # all five wide names -> wide; attn + lse -> attn; anything else -> recompute everything.
        has_attn = keep is not None and "attn" in keep and "lse" in keep
        ctx.wide = has_attn and all(n in keep for n in _WIDE_NAMES)
        ctx.kept_attn = has_attn
        if ctx.wide:
            ctx.save_for_backward(x2, ctx2, t_mod, keep["attn"], keep["lse"], *(keep[n] for n in _WIDE_NAMES))
        elif has_attn:
            ctx.save_for_backward(x2, ctx2, t_mod, keep["attn"], keep["lse"])
        else:
            ctx.save_for_backward(x2, ctx2, t_mod)
        return out

    @staticmethod
    def backward(ctx, dout):
        block, rope = ctx.block, ctx.rope
        x, c2, t_mod = ctx.saved_tensors[:3]
        kept = ctx.saved_tensors[3:5]
        wide = dict(zip(_WIDE_NAMES, ctx.saved_tensors[5:])) if ctx.wide else None
        need = dict(zip(_PARAM_NAMES, ctx.param_needs))
        any_param = any(ctx.param_needs)
        eps, dev, d = block.eps, x.device, block.dim
        sa, ca, heads, hd = block.self_attn, block.cross_attn, block.num_heads, block.dim // block.num_heads
        dout = dout.contiguous()
        g = {}                                              # parameter gradients by name

        def lin(xin, layer):
            return ops.gemm(xin, layer.weight, layer.bias)

        def lin_bwd(dy, xin, layer, prefix, want_dx=True):
            if need[prefix + ".weight"]:
                g[prefix + ".weight"] = linear_dw(dy, xin)
            if need[prefix + ".bias"]:
                g[prefix + ".bias"] = bias_grad(dy)
            return linear_dx(dy, layer.weight) if want_dx else None

        # ---- recompute the forward, un-fused, keeping the intermediates (DIT:218-229); what the forward kept (`wide`) is taken as it
        # is, and what would only feed a parameter gradient nobody asked for (a frozen block) is not computed
        mod = ops.modulation(block.modulation, t_mod.contiguous(), onep_mask=0b010010)
        need_h1 = wide is None or any(need[f"self_attn.{n}.weight"] for n in "qkv")
        h1 = ops.layernorm_modulate(x, scale1p=mod[1], shift=mod[0], eps=eps) if need_h1 else None
        if wide is not None:
            qp, kp, vv = wide["qp"], wide["kp"], wide["v"]
        else:
            qp, kp, vv = lin(h1, sa.q), lin(h1, sa.k), lin(h1, sa.v)
        qn, kn = qp.clone(), kp.clone()
        # the self-attention's q as the forward made it: pre-scaled through its rotation table, scale = ln 2 in the kernels (one
        # rounding of q, and the SAME scores in the forward's log-sum-exp and in the backward's P)
        (q_cos, q_sin), sa_scale = ((rope.scaled(Q_PRESCALE(hd)), math.log(2.0)) if ctx.q_prescale else ((rope.cos, rope.sin), None))
        ops.rmsnorm_rope(qn, sa.norm_q.weight, q_cos, q_sin, hd, sa.norm_q.eps)
        ops.rmsnorm_rope(kn, sa.norm_k.weight, rope.cos, rope.sin, hd, sa.norm_k.eps)
        a, lse = kept if len(kept) else ops.flash_attn_lse(qn, kn, vv, heads, scale=sa_scale)
        o = lin(a, sa.o) if (wide is None or need["modulation"]) else None            # o: x1 (unless kept) and the gate's gradient
        x1 = wide["x1"] if wide is not None else ops.add(x, ops.colsum(o, gate=mod[2]))   # x + gate_msa * o
        h2 = ops.layernorm_modulate(x1, weight=block.norm3.weight, bias=block.norm3.bias, eps=eps)
        q2p, k2p, v2 = lin(h2, ca.q), lin(c2, ca.k), lin(c2, ca.v)
        q2n, k2n = q2p.clone(), k2p.clone()
        ops.rmsnorm_rope(q2n, ca.norm_q.weight, None, None, hd, ca.norm_q.eps)
        ops.rmsnorm_rope(k2n, ca.norm_k.weight, None, None, hd, ca.norm_k.eps)
        a2, lse2 = ops.flash_attn_lse(q2n, k2n, v2, heads)
        x2b = wide["x2b"] if wide is not None else ops.gemm(a2, ca.o.weight, ca.o.bias, epilogue=ops.EPI_BIAS_RESID, resid=x1)
        h3 = ops.layernorm_modulate(x2b, scale1p=mod[4], shift=mod[3], eps=eps)
        u = lin(h3, block.ffn[0])
        need_y = need["modulation"]                                                      # y: only the gate's gradient reads it
        f = ops.act(u, "gelu_tanh") if (need_y or need["ffn.2.weight"]) else None
        y = lin(f, block.ffn[2]) if need_y else None

        def acc():
            return _zeros_f32(d, dev) if need["modulation"] else None

        # ---- FFN branch: x3 = x2b + gate_mlp * y
        dgate2 = acc()
        dy = ops.colsum(dout, b=y, gate=mod[5], acc=dgate2)
        df = lin_bwd(dy, f, block.ffn[2], "ffn.2")
        du = ops.act_bwd(u, df, "gelu_tanh")
        dh3 = lin_bwd(du, h3, block.ffn[0], "ffn.0")
        dscale2, dshift2 = acc(), acc()
        d_x2b = ops.add(dout, ops.layernorm_bwd(x2b, dh3, g=mod[4], dg_acc=dscale2, db_acc=dshift2, eps=eps))
        del dy, df, du, dh3, u, f, y, h3
        # ---- cross-attention branch: x2b = x1 + o2
        da2 = lin_bwd(d_x2b, a2, ca.o, "cross_attn.o")
        dq2n, dk2n, dv2 = ops.flash_attn_bwd(q2n, k2n, v2, a2, da2, lse2, heads, need_dkv=any_param)   # the context side only feeds parameter gradients
        wq_acc = _zeros_f32(d, dev) if need["cross_attn.norm_q.weight"] else None
        wk_acc = _zeros_f32(d, dev) if need["cross_attn.norm_k.weight"] else None
        dq2p = ops.rmsnorm_rope_bwd(q2p, dq2n, ca.norm_q.weight, None, None, hd, ca.norm_q.eps, dw_acc=wq_acc)
        dh2 = lin_bwd(dq2p, h2, ca.q, "cross_attn.q")
        if any_param:                                        # the context side only feeds parameter gradients
            dk2p = ops.rmsnorm_rope_bwd(k2p, dk2n, ca.norm_k.weight, None, None, hd, ca.norm_k.eps, dw_acc=wk_acc)
            lin_bwd(dk2p, c2, ca.k, "cross_attn.k", want_dx=False)
            lin_bwd(dv2, c2, ca.v, "cross_attn.v", want_dx=False)
        if wq_acc is not None:
            g["cross_attn.norm_q.weight"] = ops.f32_to_bf16(wq_acc)
        if wk_acc is not None:
            g["cross_attn.norm_k.weight"] = ops.f32_to_bf16(wk_acc)
        n3w = _zeros_f32(d, dev) if need["norm3.weight"] else None
        n3b = _zeros_f32(d, dev) if need["norm3.bias"] else None
        d_x1 = ops.add(d_x2b, ops.layernorm_bwd(x1, dh2, g=block.norm3.weight, dg_acc=n3w, db_acc=n3b, eps=eps))
        if n3w is not None:
            g["norm3.weight"] = ops.f32_to_bf16(n3w)
        if n3b is not None:
            g["norm3.bias"] = ops.f32_to_bf16(n3b)
        del da2, dq2n, dk2n, dv2, dq2p, dh2, d_x2b, a2, q2n, k2n, v2, q2p, k2p, h2, x2b
        # ---- self-attention branch: x1 = x + gate_msa * o
        dgate1 = acc()
        do = ops.colsum(d_x1, b=o, gate=mod[2], acc=dgate1)
        da = lin_bwd(do, a, sa.o, "self_attn.o")
        dqn, dkn, dvv = ops.flash_attn_bwd(qn, kn, vv, a, da, lse, heads, scale=sa_scale)
        wq_acc = _zeros_f32(d, dev) if need["self_attn.norm_q.weight"] else None
        wk_acc = _zeros_f32(d, dev) if need["self_attn.norm_k.weight"] else None
        dqp = ops.rmsnorm_rope_bwd(qp, dqn, sa.norm_q.weight, q_cos, q_sin, hd, sa.norm_q.eps, dw_acc=wq_acc)   # linear in the table: d/d(qp) of the scaled q
        dkp = ops.rmsnorm_rope_bwd(kp, dkn, sa.norm_k.weight, rope.cos, rope.sin, hd, sa.norm_k.eps, dw_acc=wk_acc)
        if wq_acc is not None:
            g["self_attn.norm_q.weight"] = ops.f32_to_bf16(wq_acc)
        if wk_acc is not None:
            g["self_attn.norm_k.weight"] = ops.f32_to_bf16(wk_acc)
        dh1 = lin_bwd(dqp, h1, sa.q, "self_attn.q")
        dh1 = ops.add(dh1, lin_bwd(dkp, h1, sa.k, "self_attn.k"))
        dh1 = ops.add(dh1, lin_bwd(dvv, h1, sa.v, "self_attn.v"))
        dscale1, dshift1 = acc(), acc()
        dx = ops.add(d_x1, ops.layernorm_bwd(x, dh1, g=mod[1], dg_acc=dscale1, db_acc=dshift1, eps=eps))
        if need["modulation"]:    # rows: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp (d(1+s) = ds)
            rows = torch.stack([dshift1, dscale1, dgate1, dshift2, dscale2, dgate2]).contiguous()
            g["modulation"] = ops.f32_to_bf16(rows).view(block.modulation.shape)
        grads = [g.get(n) if nd else None for n, nd in zip(_PARAM_NAMES, ctx.param_needs)]
        return (None, None, dx, None, None, *grads)


def block_forward(block: DiTBlock, x2, ctx2, t_mod, rope: RopeTable):
    return DiTBlockFn.apply(block, rope, x2, ctx2, t_mod, *_block_params(block))


# ---------------------------------------------------------------------------------------------------------------------
# model_fn with a tape, loss, optimiser
def model_fn_train(dit: WanModel, controlnet, latents, timestep, context, y, control_signal_video_latents):
    """model_fn_wan_video (GF:1349-1591) with gradients to the ControlNet parameters.  The DiT expert is frozen: its
    embeddings run without a tape, its blocks back-propagate activations only."""
    with torch.no_grad():
        t, t_mod = dit.time_embed(timestep)
        ctx2 = dit.embed_text(context)[0]
        x, (f, h, w) = dit.patchify(latents, extra=y if (y is not None and dit.require_vae_embedding) else None)
        rope = dit.rope_table(f, h, w, latents.device)
        pe = controlnet.controlnet_patch_embedding
        kpad = _pad64(pe.patch_embedding.weight.shape[1] * 4)
        cols = ops.patchify_im2col(control_signal_video_latents[0].contiguous(), None, kpad=kpad)
        hmod = ops.modulation(dit.head.modulation, t.contiguous(), onep_mask=0b10)
    x = x[0]
    c = PatchEmbedFn.apply(cols, pe.patch_embedding.weight, pe.patch_embedding.bias)
    n_cn = controlnet.controlnet_dit.num_layers
    for i, block in enumerate(dit.blocks):
        if i < n_cn:
            c = block_forward(controlnet.controlnet_dit.blocks[i], c, ctx2, t_mod, rope)
            zc = controlnet.controlnet_zero_convs_after[i]
            if i + 1 < n_cn:
                c, c_inj = FanoutFn.apply(c)            # the state feeds the next ControlNet block AND the injection
            else:
                c_inj = c
        x = block_forward(block, x, ctx2, t_mod, rope)
        if i < n_cn:
            x = InjectFn.apply(x, c_inj, zc.weight.view(zc.weight.shape[0], -1), zc.bias)
    hx = LayerNormModFn.apply(x, hmod[1], hmod[0], dit.head.eps)
    tok = LinearFn.apply(hx, dit.head.head.weight, dit.head.head.bias)
    return UnpatchifyFn.apply(tok, dit.out_dim, f, h, w).unsqueeze(0)


def training_loss(pipe, *, input_latents, noise, context, y, control_signal_video_latents, timestep_id=None,
                  max_timestep_boundary=1.0, min_timestep_boundary=0.0, dit=None, controlnet=None):
    """GF:180-193.  The scheduler must be in training mode (`set_timesteps(1000, training=True)`, as the reference's
    training entry point does).  `timestep_id` pins the random draw (tests)."""
    sch = pipe.scheduler
    if timestep_id is None:
        lo = int(min_timestep_boundary * sch.num_train_timesteps)
        hi = int(max_timestep_boundary * sch.num_train_timesteps)
        timestep_id = int(torch.randint(lo, hi, (1,)))
    timestep = sch.timesteps[timestep_id:timestep_id + 1].to(dtype=pipe.torch_dtype, device=pipe.device)
    latents = sch.add_noise(input_latents, noise, timestep)
    target = sch.training_target(input_latents, noise, timestep)
    pred = model_fn_train(dit or pipe.dit, controlnet or pipe.controlnet, latents, timestep, context, y,
                          control_signal_video_latents)
    return MseLossFn.apply(pred[0], target[0].contiguous(), float(sch.training_weight(timestep)))


def forward_preprocess(pipe, data, extra_inputs=("input_image",), tiled=False, tile_size=(30, 52), tile_stride=(15, 26)):
    """`WanTrainingModule.forward_preprocess` (scripts/train/train.py:76-118): one dataset item — {"video": list of PIL frames,
    "prompt": str, "control_video": [F,H,W,3] tensor} — through the pipeline's units in TRAINING mode, i.e. what the reference's
    loop `for unit in self.pipe.units: ...` produces for the inputs Goal Force trains on (cfg_scale 1: no negative prompt, UTIL:262-271;
    tiled False; extra_inputs "input_image" = the clip's first frame, train.py:104-106):
        ShapeChecker GF:741-747 · NoiseInitializer GF:751-763 (unseeded: `rand_device = pipe.device`, train.py:93) ·
        PromptEmbedder GF:808-820 · InputVideoEmbedder GF:767-789 (training branch: latents = noise, input_latents = vae.encode(video)) ·
        ControlVideoEmbedder GF:791-805 · ImageEmbedderVAE GF:887-917.
    Returns the keyword arguments of training_loss: input_latents, noise, context, y, control_signal_video_latents."""
    if not pipe.scheduler.training:
        raise GoalForceError("forward_preprocess: put the scheduler in training mode first (set_timesteps(1000, training=True), utils.py:560)")
    video = data["video"]
    width, height = video[0].size
    num_frames = len(video)
    h2, w2, f2 = pipe.check_resize_height_width(height, width, num_frames)
    if (h2, w2, f2) != (height, width, num_frames):
        raise GoalForceError(f"forward_preprocess: a clip of {num_frames} frames of {width}x{height} does not fit the model's grid "
                             f"({f2} x {w2}x{h2}): the reference would encode the clip as it is and fail in the patch embedding")
    noise = pipe.generate_noise((1, 16, (num_frames - 1) // 4 + 1, height // 8, width // 8), seed=None, rand_device=pipe.device)
    pr = pipe.prompter
    from .text_encoder import WanPrompter
    if isinstance(pr, WanPrompter):
        if pipe.text_encoder is None or pr.tokenizer is None:
            raise GoalForceError("forward_preprocess: no text encoder / tokenizer loaded")
        pr.fetch_models(pipe.text_encoder)
    context = pr.encode_prompt(data["prompt"], positive=None, device=pipe.device)          # GF:812-813: `positive` is never in the inputs
    input_latents = pipe.embed_input_video(video, tiled, tile_size, tile_stride).to(dtype=pipe.torch_dtype)
    control = pipe.embed_control_video(data["control_video"], tiled, tile_size, tile_stride)
    y = None
    if "input_image" in extra_inputs:
        y = pipe.embed_image(video[0], num_frames, height, width, tiled, tile_size, tile_stride)
    unknown = [e for e in extra_inputs if e != "input_image"]
    if unknown:
        raise NotImplementedError(f"extra_inputs {unknown}: pipeline branches Goal Force never trains (train.py:107-112)")
    return dict(input_latents=input_latents, noise=noise, context=context, y=y, control_signal_video_latents=control)


class AdamW:
    """torch.optim.AdamW(params, lr, weight_decay) of launch_training_task (utils.py:755) on the HIP kernel: bf16
    parameters, fp32 moments (the reference's moments inherit the parameters' bf16; fp32 is strictly more precise)."""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def grad_norm(self) -> float:
        sq = torch.zeros((1,), dtype=torch.float32, device=self.params[0].device)
        for p in self.params:
            if p.grad is not None:
                ops.sumsq(p.grad.contiguous(), sq)
        return math.sqrt(float(sq[0]))

    @torch.no_grad()
    def step(self, max_grad_norm: Optional[float] = None, grad_norm: Optional[float] = None):
        """One update; `max_grad_norm` > 0 applies accelerator.clip_grad_norm_ (utils.py:806-808) as a gradient scale
        (`grad_norm`: the norm when the caller has already computed it)."""
        self.step_count += 1
        scale = 1.0
        if max_grad_norm is not None and max_grad_norm > 0:
            n = self.grad_norm() if grad_norm is None else grad_norm
            scale = min(1.0, max_grad_norm / (n + 1e-6))
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = (torch.zeros(p.shape, dtype=torch.float32, device=p.device),
                                      torch.zeros(p.shape, dtype=torch.float32, device=p.device))
            ops.adamw_step(p.detach(), p.grad.contiguous(), st[0], st[1], self.step_count, self.lr, self.betas, self.eps,
                           self.weight_decay, grad_scale=scale)


def allreduce_gradients(params, group=None, bucket_bytes: int = 512 << 20):
    """Data-parallel gradient averaging (the reference trains under Accelerate's DistributedDataParallel,
    utils.py:756-760): gradients are packed into flat bf16 buckets and all-reduced over RCCL.  xGMI is point to point, so a
    ring all-reduce moves 2(N-1)/N of each bucket over every link; 512 MB buckets (the ControlNet has 7.5 GB of bf16
    gradients -> 15 collectives) keep the per-collective launch latency negligible against the ~25 ms a bucket takes at
    ~40 GB/s per direction and link pair, and the whole exchange (~0.4 s) is small against the 9 s backward it follows.
    Called after backward(), before the optimiser step; every rank must hold the same set of gradients."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.requires_grad and p.grad is not None]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat = (flat.float() / world).to(bucket[0].dtype)          # mean, rounded once
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


class ConstantLR:
    """`torch.optim.lr_scheduler.ConstantLR(optimizer)` as launch_training_task builds it (utils.py:756) — with torch's DEFAULTS, i.e.
    the learning rate is lr / 3 for the first 5 optimiser steps and lr from the sixth on (factor 1/3, total_iters 5).  The same
    chained arithmetic as torch (`lr * factor`, later `lr * (1 / factor)`), so the floats agree."""

    def __init__(self, optimizer: "AdamW", factor: float = 1.0 / 3, total_iters: int = 5):
        self.optimizer, self.factor, self.total_iters, self.last_epoch = optimizer, factor, total_iters, 0
        optimizer.lr = optimizer.lr * factor

    def step(self):
        self.last_epoch += 1
        if self.last_epoch == self.total_iters:
            self.optimizer.lr = self.optimizer.lr * (1.0 / self.factor)


class ModelLogger:
    """utils.py:592-650: counts optimiser steps, writes `step-<N>.safetensors` every `save_steps` (and once more at the end if the last
    step was not a multiple), or `epoch-<k>.safetensors` when save_steps is None.  Checkpoint keys: the trainable parameters under
    their names in the training module, `pipe.controlnet.<name>` (the scripts pass --remove_prefix_in_ckpt "pipe.dit.", which leaves
    them as they are; `load_controlnet_weights` strips `pipe.controlnet.`, GF:176-178).  Written by rank 0 only."""

    def __init__(self, output_path, remove_prefix_in_ckpt=None, log=None, log_every=10):
        self.output_path, self.remove_prefix_in_ckpt, self.num_steps = output_path, remove_prefix_in_ckpt, 0
        self.log, self.log_every = log, log_every

    def on_step_end(self, pipe, loss, learning_rate, grad_norm=None, save_steps=None):
        self.num_steps += 1
        if self.log is not None and self.num_steps % self.log_every == 0:            # wandb.log every 10 steps (utils.py:604-615)
            rec = {"training_loss": float(loss), "learning_rate": learning_rate}
            if grad_norm is not None:
                rec["gradient_norm"] = grad_norm
            self.log(rec, self.num_steps)
        if save_steps is not None and self.num_steps % save_steps == 0:
            self.save_model(pipe, f"step-{self.num_steps}.safetensors")

    def on_epoch_end(self, pipe, epoch_id):
        self.save_model(pipe, f"epoch-{epoch_id}.safetensors")

    def on_training_end(self, pipe, save_steps=None):
        if save_steps is not None and self.num_steps % save_steps != 0:
            self.save_model(pipe, f"step-{self.num_steps}.safetensors")

    def save_model(self, pipe, file_name):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            if dist.get_rank() != 0:
                return None
        from safetensors.torch import save_file
        sd = {"pipe.controlnet." + k: v.detach().cpu().contiguous() for k, v in pipe.controlnet.named_parameters() if v.requires_grad}
        if self.remove_prefix_in_ckpt is not None:
            sd = {(k[len(self.remove_prefix_in_ckpt):] if k.startswith(self.remove_prefix_in_ckpt) else k): v for k, v in sd.items()}
        os.makedirs(self.output_path, exist_ok=True)
        path = os.path.join(self.output_path, file_name)
        save_file(sd, path)
        return path


def data_is_correct_shape_and_type(data, control_signal_type, num_frames) -> bool:
    """utils.py:653-679: every frame a PIL image of 832 x 480, the control video [num_frames, 480, 832, 3]."""
    from PIL import Image
    if control_signal_type not in ("canny_edge", "direct_force_and_goal_force_and_mass"):
        raise NotImplementedError(control_signal_type)
    ok = all(isinstance(f, Image.Image) and f.size == (832, 480) for f in data["video"])
    return ok and tuple(data["control_video"].shape) == (num_frames, 480, 832, 3)


def should_skip_batch(local_condition_is_met: bool, device="cpu") -> bool:
    """utils.py:682-698 — the ONE collective of the reference's own code: any rank with a bad batch makes every rank skip it."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(local_condition_is_met)
    t = torch.tensor(1.0 if local_condition_is_met else 0.0, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t) > 0


def safe_collate(batch):
    """utils.py:700-715: drop the samples that failed to load; the first of what is left (batch size 1), None if nothing is."""
    batch = [b for b in batch if b is not None]
    return batch[0] if batch else None


def launch_training_task(dataset, model, learning_rate: float = 1e-5, weight_decay: float = 1e-2, num_workers: int = 8,
                         save_steps: Optional[int] = None, num_epochs: int = 1, gradient_accumulation_steps: int = 1,
                         find_unused_parameters: bool = False, args=None, forward=None, shuffle: bool = True, log=None, now=None):
    """`launch_training_task` (utils.py:734-826) on the HIP training step: AdamW over the trainable (ControlNet) parameters,
    `ConstantLR` (lr / 3 for the first five steps), a shuffled DataLoader with `safe_collate`, the bad-batch consensus, gradient
    clipping at `args.max_grad_norm` (> -1), `ModelLogger` checkpoints, and the resume rule — a `--controlnet_checkpoint
    .../step-<N>.safetensors` moves the output directory next to it, fast-forwards the LR schedule N steps and continues the step
    count at N + 1 (sic, utils.py:781-785).  Data parallel: one process per GPU under torch.distributed (RCCL); gradients are
    averaged by `allreduce_gradients` where the reference uses Accelerate's DDP.
    `model`: the pipeline, or an object with `.pipe` (+ optional extra_inputs / max_timestep_boundary / min_timestep_boundary, as
    WanTrainingModule carries them, train.py:70-74).  `forward(pipe, data) -> loss` replaces the default
    `training_loss(pipe, **forward_preprocess(pipe, data))` (tests pin the random draws through it); `shuffle`, `log(record, step)`
    (where the reference calls wandb.log) and `now` (the run directory's time stamp) are the other seams.  Returns the ModelLogger."""
    import datetime
    if args is not None:
        learning_rate, weight_decay, num_workers = args.learning_rate, args.weight_decay, args.dataset_num_workers
        save_steps, num_epochs = args.save_steps, args.num_epochs
        gradient_accumulation_steps = args.gradient_accumulation_steps
    if gradient_accumulation_steps != 1:
        raise NotImplementedError("gradient_accumulation_steps > 1 (every script of the reference trains with 1)")
    pipe = getattr(model, "pipe", model)
    extra = tuple(getattr(model, "extra_inputs", None) or ("input_image",))
    tmax, tmin = getattr(model, "max_timestep_boundary", 1.0), getattr(model, "min_timestep_boundary", 0.0)
    if forward is None:
        def forward(pipe_, data):
            return training_loss(pipe_, **forward_preprocess(pipe_, data, extra_inputs=extra), max_timestep_boundary=tmax,
                                 min_timestep_boundary=tmin)
    params = [p_ for p_ in pipe.controlnet.parameters() if p_.requires_grad]
    optimizer = AdamW(params, lr=learning_rate, weight_decay=weight_decay)
    scheduler = ConstantLR(optimizer)
    print("Dataset size: ", len(dataset))
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    sampler = None
    if world > 1:
        # what accelerator.prepare(dataloader) does to the reference's loader: every process draws its share of ONE shuffled order per
        # epoch; the shares are padded to equal length, so every rank runs the same number of steps (the bad-batch consensus and the
        # gradient all-reduce are collectives: unequal step counts would hang)
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=world, rank=dist.get_rank(), shuffle=shuffle, drop_last=False)
    dataloader = torch.utils.data.DataLoader(dataset, shuffle=shuffle if sampler is None else False, sampler=sampler, collate_fn=safe_collate,
                                             num_workers=num_workers)
    ckpt = getattr(args, "controlnet_checkpoint", None)
    output_path = getattr(args, "output_path", "./models")
    if ckpt is None:
        stamp = (now or datetime.datetime.now()).strftime("%Y-%m-%d_%H-%M-%S")                 # utils.py:18-26, 768-771
        output_path = os.path.join(output_path, stamp)
    else:
        output_path = os.path.dirname(ckpt)                                                    # utils.py:773-775
    logger = ModelLogger(output_path, remove_prefix_in_ckpt=getattr(args, "remove_prefix_in_ckpt", None), log=log)
    if ckpt is not None:
        step_num_initial = int(ckpt.split("/")[-1].split("step-")[1].split(".")[0])
        for _ in range(step_num_initial):
            scheduler.step()
        logger.num_steps = step_num_initial + 1
    max_grad_norm = getattr(args, "max_grad_norm", -1)
    for epoch_id in range(num_epochs):
        if sampler is not None:
            sampler.set_epoch(epoch_id)
        for data in dataloader:
            bad = data is None or not data_is_correct_shape_and_type(data, args.control_signal_type, args.num_frames)
            if should_skip_batch(bad, pipe.device):
                print("--> A bad batch was detected across GPUs. Skipping. <--")
                continue
            optimizer.zero_grad()
            with torch.enable_grad():            # whatever the caller's global grad mode is (inference code switches it off process-wide)
                loss = forward(pipe, data)
            if not math.isfinite(float(loss.detach())):
                # the reference would step the optimiser on NaN gradients and write NaN checkpoints from here on; stop with the item's name
                raise GoalForceError(f"launch_training_task: non-finite loss {float(loss.detach())} at step {logger.num_steps + 1} "
                                     f"(item {data.get('file_id', '?')!r}): no optimiser step taken, nothing written")
            loss.backward()
            allreduce_gradients(params)
            grad_norm = optimizer.grad_norm() if max_grad_norm > -1 else None                   # clip_grad_norm_ returns the norm BEFORE clipping
            optimizer.step(max_grad_norm=max_grad_norm if max_grad_norm > -1 else None, grad_norm=grad_norm)
            logger.on_step_end(pipe, loss.detach(), optimizer.lr, grad_norm=grad_norm, save_steps=save_steps)
            # Accelerate's prepared scheduler advances the wrapped one `num_processes` times per training step (accelerate/scheduler.py,
            # AcceleratedScheduler.step without split_batches): on the reference's 4-process runs the lr / 3 phase lasts 2 steps, not 5
            for _ in range(world):
                scheduler.step()
        if save_steps is None:
            logger.on_epoch_end(pipe, epoch_id)
    logger.on_training_end(pipe, save_steps)
    return logger


def wan_parser():
    """`wan_parser` (utils.py:854-900): the training script's command line, argument for argument and default for default (pinned
    against the reference's own parser object by tests/golden/g18).  Arguments of branches Goal Force never trains (LoRA, strided
    ControlNet, gradient-checkpoint offload, wandb) are accepted so that the reference's shell scripts run unchanged;
    scripts/train.py refuses the ones that would change what is computed."""
    import argparse
    ap = argparse.ArgumentParser(description="Simple example of a training script.")
    ap.add_argument("--dataset_base_path", type=str, nargs="+", default="", required=True, help="Base path(s) of the dataset.")
    ap.add_argument("--dataset_metadata_path", type=str, nargs="+", default=None, help="Path(s) to the metadata file of the dataset.")
    ap.add_argument("--controlnet_checkpoint", type=str, default=None, help="Path to the ckpt file of the previously trained controlnet.")
    ap.add_argument("--control_signal_type", type=str, default=None, help="Type of control signal.")
    ap.add_argument("--controlnet_num_layers", type=int, default=0, help="Number of DiT layers for the ControlNet")
    ap.add_argument("--controlnet_stride", type=int, default=None)
    ap.add_argument("--apply_strided_controlnet", action="store_true")
    ap.add_argument("--offline_load", action="store_true", help="Whether to load models from offline local model files.")
    ap.add_argument("--max_pixels", type=int, default=1280 * 720)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--num_frames", type=int, default=81)
    ap.add_argument("--data_file_keys", type=str, default="image,video")
    ap.add_argument("--dataset_repeat", type=int, default=1)
    ap.add_argument("--model_paths", type=str, default=None, help="Paths to load models. In JSON format.")
    ap.add_argument("--model_id_with_origin_paths", type=str, default=None)
    ap.add_argument("--learning_rate", type=float, default=1e-4)
    ap.add_argument("--num_epochs", type=int, default=1)
    ap.add_argument("--output_path", type=str, default="./models")
    ap.add_argument("--remove_prefix_in_ckpt", type=str, default="pipe.dit.")
    ap.add_argument("--trainable_models", type=str, default=None)
    ap.add_argument("--lora_base_model", type=str, default=None)
    ap.add_argument("--lora_target_modules", type=str, default="q,k,v,o,ffn.0,ffn.2")
    ap.add_argument("--lora_rank", type=int, default=32)
    ap.add_argument("--lora_checkpoint", type=str, default=None)
    ap.add_argument("--extra_inputs", default=None, help="Additional model inputs, comma-separated.")
    ap.add_argument("--use_gradient_checkpointing_offload", default=False, action="store_true")
    ap.add_argument("--gradient_accumulation_steps", type=int, default=1)
    ap.add_argument("--max_timestep_boundary", type=float, default=1.0)
    ap.add_argument("--min_timestep_boundary", type=float, default=0.0)
    ap.add_argument("--find_unused_parameters", default=False, action="store_true")
    ap.add_argument("--save_steps", type=int, default=None)
    ap.add_argument("--dataset_num_workers", type=int, default=0)
    ap.add_argument("--weight_decay", type=float, default=0.01)
    ap.add_argument("--wandb_logging", default=False, action="store_true")
    ap.add_argument("--wandb_project", type=str, default="diffsynth-training")
    ap.add_argument("--wandb_run_name", type=str, default=None)
    ap.add_argument("--max_grad_norm", type=float, default=-1)
    ap.add_argument("--p_mask_out_direct_force", type=float, default=0.0)
    ap.add_argument("--p_mask_out_indirect_force", type=float, default=0.0)
    ap.add_argument("--p_mask_out_masses", type=float, default=0.0)
    return ap


def get_dataset(args, device="cuda", video_loader=None):
    """`get_dataset` (scripts/train/train.py:126-197): `canny_edge` -> the Canny set on its own video operator; 
    `direct_force_and_goal_force_and_mass` -> ConcatDataset([balls, dominos, plants]) in TRAINING mode, base / metadata paths in that
    order, the three masking probabilities on the first two."""
    from .canny import ControlSignalDataset_CannyEdge
    from .force_map import ControlSignalDataset_Balls, ControlSignalDataset_Dominos, ControlSignalDataset_Plants
    if args.control_signal_type == "canny_edge":
        return ControlSignalDataset_CannyEdge(
            base_path=args.dataset_base_path[0], metadata_path=args.dataset_metadata_path[0], repeat=args.dataset_repeat,
            data_file_keys=args.data_file_keys.split(","), device=device,
            main_data_operator=ControlSignalDataset_CannyEdge.default_video_operator(
                base_path=args.dataset_base_path[0], max_pixels=args.max_pixels, height=args.height, width=args.width,
                height_division_factor=16, width_division_factor=16, num_frames=args.num_frames, time_division_factor=4,
                time_division_remainder=1))
    if args.control_signal_type != "direct_force_and_goal_force_and_mass":
        raise NotImplementedError(args.control_signal_type)
    common = dict(repeat=args.dataset_repeat, data_file_keys=args.data_file_keys.split(","), is_validation_dataset=False,
                  num_frames=args.num_frames, height=args.height, width=args.width, device=device, video_loader=video_loader)
    masks = dict(p_mask_out_masses=args.p_mask_out_masses, p_mask_out_direct_force=args.p_mask_out_direct_force,
                 p_mask_out_indirect_force=args.p_mask_out_indirect_force)
    balls = ControlSignalDataset_Balls(base_path=args.dataset_base_path[0], metadata_path=args.dataset_metadata_path[0], **common, **masks)
    dominos = ControlSignalDataset_Dominos(base_path=args.dataset_base_path[1], metadata_path=args.dataset_metadata_path[1], **common, **masks)
    plants = ControlSignalDataset_Plants(base_path=args.dataset_base_path[2], metadata_path=args.dataset_metadata_path[2], **common)
    return torch.utils.data.ConcatDataset([balls, dominos, plants])


def controlnet_state_dict(controlnet: nn.Module) -> dict:
    """Checkpoint layout of the reference's ModelLogger (remove_prefix 'pipe.controlnet.' on load, GF:176-178)."""
    return {"pipe.controlnet." + k: v.detach().cpu() for k, v in controlnet.state_dict().items()}
