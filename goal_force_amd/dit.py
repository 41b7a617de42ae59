"""Wan DiT (Wan2.2-I2V-A14B expert) — host-side mirror of the reference's module interface with
every forward implemented by the HIP kernels in libgoalforce_hip.so.

Drop-in surface: class names, constructor arguments, forward signatures and
state_dict key names follow diffsynth/models/wan_video_dit.py (DIT) so reference checkpoints load
unchanged (`blocks.N.self_attn.q.weight`, ..., DIT:273-340).  The nn.Linear / nn.LayerNorm
submodules are parameter containers only — their torch forwards are never called.

Activations are token-major [S, D] bf16 (batch 1 per call on the hot path; B > 1 is looped).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import GoalForceError


# ------------------------------------------------------------------ RoPE tables (host, fp64 -> fp32)
def precompute_freqs_cis(dim: int, end: int = 1024, theta: float = 10000.0):
    """DIT:83-89 — complex128 table [end, dim/2]."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].double() / dim))
    freqs = torch.outer(torch.arange(end), freqs)
    return torch.polar(torch.ones_like(freqs), freqs)


def precompute_freqs_cis_3d(dim: int, end: int = 1024, theta: float = 10000.0):
    """DIT:75-80 — (frame, height, width) tables; pairs split (d-2(d//3))/2, (d//3)/2, (d//3)/2."""
    return (precompute_freqs_cis(dim - 2 * (dim // 3), end, theta),
            precompute_freqs_cis(dim // 3, end, theta),
            precompute_freqs_cis(dim // 3, end, theta))


class RopeTable:
    """cos/sin of the per-token rotary phases, resident in HBM as fp32 [S, head_dim/2] each.
    Built once per (f, h, w) grid instead of the reference's per-forward CPU rebuild + H2D copy
    (GF:1474-1478)."""

    def __init__(self, freqs_complex: torch.Tensor, device):
        fc = freqs_complex.reshape(freqs_complex.shape[0], -1)
        self.cos = fc.real.to(torch.float32).contiguous().to(device)
        self.sin = fc.imag.to(torch.float32).contiguous().to(device)
        self.tokens = fc.shape[0]
        self._scaled = {}

    def scaled(self, c: float):
        """(c cos, c sin): the rotation that also multiplies by c.  The self-attention's Q goes through it with c = softmax scale x
        log2(e) (SelfAttention.attend), so that the attention kernel's operand Q' = bf16(c q) is produced from the fp32 rotation with
        ONE rounding — where rounding q to bf16 first and c q again (what the kernel does to a plain q) costs the logits a second one."""
        tabs = self.__dict__.setdefault("_scaled", {})          # (shard tables are built without __init__: sequence_parallel.shard_rope)
        if c not in tabs:
            tabs[c] = ((self.cos * c).contiguous(), (self.sin * c).contiguous())
        return tabs[c]

    @staticmethod
    def from_grid(freqs3: Tuple[torch.Tensor, torch.Tensor, torch.Tensor], f: int, h: int, w: int, device):
        tab = torch.cat([
            freqs3[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
            freqs3[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
            freqs3[2][:w].view(1, 1, w, -1).expand(f, h, w, -1),
        ], dim=-1).reshape(f * h * w, -1)
        return RopeTable(tab, device)


def Q_PRESCALE(head_dim: int) -> float:
    """c = fp32(fp32(1 / sqrt(head_dim)) x fp32(log2 e)) — bit for bit the factor the attention launcher folds into Q (gf_attention.hip:
    `scale_log2e = scale * 1.4426950408889634f`)."""
    import numpy as np
    return float(np.float32(np.float32(1.0 / math.sqrt(head_dim)) * np.float32(1.4426950408889634)))


def _as_rope(freqs, device) -> RopeTable:
    if isinstance(freqs, RopeTable):
        return freqs
    if torch.is_tensor(freqs) and freqs.is_complex():  # the reference's [S,1,d/2] complex table
        return RopeTable(freqs.detach().cpu(), device)
    raise GoalForceError("freqs must be a RopeTable or the reference's complex [S,1,d/2] tensor")


_SINUSOID_FREQS = {}


def sinusoidal_embedding_1d(dim, position):
    """DIT:68-72: cat(cos, sin)(t * 10000^(-i/(dim/2))) in fp64, cast to position.dtype.  The frequency table is built once
    on the host (the reference's fp64 `pow`) and kept on the timestep's device; the product, cos and sin run there in
    fp64 too, so a forward no longer pulls the timestep to the host (a hidden D2H sync, 100 x per video)."""
    key = (dim, position.device)
    inv = _SINUSOID_FREQS.get(key)
    if inv is None:
        inv = torch.pow(10000, -torch.arange(dim // 2, dtype=torch.float64).div(dim // 2)).to(position.device)
        _SINUSOID_FREQS[key] = inv
    sinusoid = torch.outer(position.detach().type(torch.float64), inv)
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1).to(position.dtype)


# ------------------------------------------------------------------ B3: the reference's module-level functions (DIT:28, 64, 92)
def flash_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, compatibility_mode=False):
    """DIT:28-61 — q, k, v [B, S, num_heads * head_dim] (`b s (n d)`) -> [B, Sq, num_heads * head_dim]: softmax(q k^T / sqrt(d)) v
    per head, no mask, on the HIP flash-attention kernels (gf_flash_attn_fwd).  `compatibility_mode` only chooses among the
    reference's backends (flash-attn 3 / 2, sageattention, torch SDPA); there is one backend here, so it is accepted and ignored.
    The q handed in is final (already rounded to bf16 by its producer), so long key sequences run on the kernel that scales the fp32
    scores — SDPA's precision at peaky logits — not on the one that pre-scales and re-rounds Q (ops.flash_attn: finished_q); the
    package's own blocks produce a pre-scaled q instead and keep the faster kernel (SelfAttention.attend)."""
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3 or not (q.shape[0] == k.shape[0] == v.shape[0]):
        raise GoalForceError("flash_attention: q, k, v must be [B, S, num_heads * head_dim] with equal B")
    return torch.stack([ops.flash_attn(q[b].contiguous(), k[b].contiguous(), v[b].contiguous(), num_heads, finished_q=True)
                        for b in range(q.shape[0])])


def modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor):
    """DIT:64-65 — x * (1 + scale) + shift with x [B, S, D] and shift / scale broadcastable [B or 1, 1, D] (or [D]), computed in
    the reference's eager bf16 rounding sequence by gf_modulate."""
    if x.dim() == 2:
        return ops.modulate(x, shift.reshape(-1).contiguous(), scale.reshape(-1).contiguous())
    d = x.shape[-1]
    sh, sc = (t.reshape(-1, d) if t.numel() != d else t.reshape(1, d) for t in (shift, scale))
    if sh.shape[0] not in (1, x.shape[0]) or sc.shape[0] not in (1, x.shape[0]):
        raise GoalForceError("modulate: shift / scale must be [B or 1, 1, D] (per-token modulation is outside the Goal-Force path)")
    return torch.stack([ops.modulate(x[b].contiguous(), sh[b if sh.shape[0] > 1 else 0].contiguous(),
                                     sc[b if sc.shape[0] > 1 else 0].contiguous()) for b in range(x.shape[0])])


def rope_apply(x: torch.Tensor, freqs, num_heads: int):
    """DIT:92-97 — x [B, S, num_heads * head_dim], freqs the reference's complex [S, 1, head_dim / 2] table (or a RopeTable):
    adjacent pairs of every head rotated by the token's phases (gf_rope_apply, fp32 math, one rounding to bf16)."""
    rope = _as_rope(freqs, x.device)
    hd = x.shape[-1] // num_heads
    return torch.stack([ops.rope_apply(x[b].contiguous(), rope.cos, rope.sin, hd) for b in range(x.shape[0])])


_CACHE_EPOCH = [0]


def invalidate_caches():
    """Every copy derived from a parameter (K-padded patch weights, fp8 weight copies, the ControlNet's all-zero flag, memoised
    context projections) is rebuilt on its next use.  Needed only for changes param_key cannot see: a raw-pointer write to a
    weight, or an in-place edit of an inference tensor."""
    _CACHE_EPOCH[0] += 1


def param_key(*tensors):
    """Identity + modification state of parameters a derived copy was built from: (object id, device, storage address, autograd
    version) per tensor.  Optimiser steps (ops.adamw_step bumps the version), load_state_dict, .to() and in-place edits all change
    it, so a cache that compares keys can never serve weights captured before an update.  Inference tensors (a pipeline built
    under torch.inference_mode()) carry no version counter — reading `_version` raises there — so their key is identity +
    address only: after an in-place edit of such a weight (or a raw-pointer write to any weight) call `invalidate_caches()`,
    the hook for changes the key cannot see."""
    return (_CACHE_EPOCH[0],) + tuple((id(t), str(t.device), t.data_ptr(), None if t.is_inference() else t._version) for t in tensors)


class QuantizedInput:
    """Row-quantised activation (e4m3 bytes + per-row scale) shared by the GEMMs that read the same input."""

    def __init__(self, x2=None, x8=None, scale=None):
        self.x8, self.scale = ops.quant_fp8_rowscale(x2) if x2 is not None else (x8, scale)

    @classmethod
    def layernorm(cls, x2, **kw):
        """LayerNorm(+affine)(+modulate) of x2 delivered as an fp8_linear input (ops.layernorm_modulate_fp8)."""
        return cls(None, *ops.layernorm_modulate_fp8(x2, **kw))

    @property
    def shape(self):
        return self.x8.shape

    @property
    def is_cuda(self):
        return self.x8.is_cuda


def linear(x2, lin: nn.Linear, epilogue=ops.EPI_BIAS, resid=None, gate=None, out=None):
    """One nn.Linear on the MFMA GEMM.  If the layer carries an fp8 weight copy (enable_fp8) the reference's
    fp8_linear contract is used (VRAM:115-151: per-row dynamic activation scale, unit weight scale, e4m3),
    otherwise the bf16 kernel.  x2 may be a QuantizedInput to reuse one quantisation for several layers."""
    w8 = getattr(lin, "_gf_w8", None)
    if w8 is None:
        return ops.gemm(x2, lin.weight, lin.bias, epilogue=epilogue, resid=resid, gate=gate, out=out)
    if lin._gf_w8_key != param_key(lin.weight):          # the bf16 master changed since the cast: refresh the copy
        w8 = lin._gf_w8 = ops.cast_fp8(lin.weight.detach().contiguous())
        lin._gf_w8_key = param_key(lin.weight)
    q = x2 if isinstance(x2, QuantizedInput) else QuantizedInput(x2)
    return ops.gemm_fp8(q.x8, q.scale, w8, lin.bias, epilogue=epilogue, resid=resid, gate=gate, out=out)


def enable_fp8(module: nn.Module, enabled=True):
    """BASELINE config 5: every nn.Linear inside the DiT / ControlNet *blocks* runs the fp8_linear contract
    (the reference would set computation_dtype=float8_e4m3fn in enable_vram_management, VRAM:113).  Weights are
    cast once to OCP e4m3 (unit scale) and kept beside the bf16 master copy."""
    for blk in module.modules():
        if isinstance(blk, DiTBlock):
            for lin in blk.modules():
                if isinstance(lin, nn.Linear):
                    lin._gf_w8 = ops.cast_fp8(lin.weight.detach().contiguous()) if enabled else None
                    lin._gf_w8_key = param_key(lin.weight) if enabled else None
    return module



def pad_run(ctx2: torch.Tensor) -> int:
    """First row index n of the run of identical rows that ends the [L, D] tensor (rows n .. L-1 are all equal): L - 1 when the
    last two rows differ.  One comparison kernel and one 8-byte read-back per call.  The run is a property of the embedded context,
    so model_fn_wan_video calls this ONCE per embedded context (ContextCache.pad_n) and hands n to every block."""
    L = ctx2.shape[0]
    if L < 2:
        return max(L - 1, 0)
    differs = (ctx2[1:] != ctx2[:-1]).any(dim=1)                 # differs[i]: row i + 1 != row i
    idx = torch.nonzero(differs)
    return int(idx[-1]) + 1 if idx.numel() else 0


def _tokens2d(x: torch.Tensor) -> torch.Tensor:
    """[1,S,D] or [S,D] -> [S,D] view."""
    if x.dim() == 3:
        if x.shape[0] != 1:
            raise GoalForceError("hot-path modules take batch 1 ([1,S,D]); loop over the batch")
        return x[0]
    return x


# ------------------------------------------------------------------ modules
class RMSNorm(nn.Module):
    """DIT:100-111 (full-width, fp32 math, bf16 weight multiply)."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        y = x.clone()
        ops.rmsnorm_rope(_tokens2d(y), self.weight, None, None, head_dim=8, eps=self.eps)
        return y


class SelfAttention(nn.Module):
    """DIT:124-147."""

    def __init__(self, dim: int, num_heads: int, eps: float = 1e-6):
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = RMSNorm(dim, eps=eps), RMSNorm(dim, eps=eps)

    def attend(self, x2: torch.Tensor, rope: RopeTable, sp=None, keep=None) -> torch.Tensor:
        """x2 [S,D] -> attention output BEFORE the o projection, [S,D].  With a sequence_parallel.SequenceParallel
        group `sp`, x2 and rope are this rank's token chunk and the heads are exchanged over xGMI (usp_attn_forward,
        diffsynth/distributed/xdit_context_parallel.py:109-131).  `keep` (a dict, training): the attention output and the
        rows' log-sum-exp are stored under "attn" / "lse" so that the backward does not run the attention again; with
        keep["wide"] also the three projections ("qp", "kp" before their norm, "v")."""
        fp8 = getattr(self.q, "_gf_w8", None) is not None
        xin = (x2 if isinstance(x2, QuantizedInput) else QuantizedInput(x2)) if fp8 else x2
        # Q leaves its RMSNorm + RoPE kernel already multiplied by c = softmax scale x log2(e) (the rotation table carries the factor:
        # RopeTable.scaled), and the attention is called with scale = ln 2, i.e. c = 1 inside — its `Q <- bf16(Q c)` is then exact.
        # One rounding of the rotated q instead of two: at peaky logits (std 3) the self-attention kernel is 3.9e-3 from fp64 on a
        # twice-rounded Q and at torch SDPA's level (1.8e-3) on this one (tools/fuzz_ops.py, profiles/r06).  Training too (`keep`):
        # the backward rebuilds P from q.k and the forward's log-sum-exp, and a forward that rounded Q a second time hands it an lse
        # of OTHER scores — rows of P off by 2^-9 of the logit, dQ / dK / dV 1.7e-2 from fp64 at logit std 8 where torch's SDPA
        # backward is at 4e-3 (tools/attnbwd_precision.py).  On the pre-scaled q forward and backward see the same scores; the
        # backward of the norm + rotation takes the same scaled table (training.DiTBlockFn) and so returns d/d(projection).
        q_cos, q_sin, attn_scale = rope.cos, rope.sin, None
        if ops._OPT["attn_q_prescale"]:
            (q_cos, q_sin), attn_scale = rope.scaled(Q_PRESCALE(self.head_dim)), math.log(2.0)
        if sp is not None and keep is None:
            # head-parallel: every projection's tokens-for-heads exchange starts as soon as the projection is ready and flies under
            # the next one (same kernels on the same values as below: bit-identical, another issue order)
            k = linear(xin, self.k)
            ops.rmsnorm_rope(k, self.norm_k.weight, rope.cos, rope.sin, self.head_dim, self.norm_k.eps)
            hk = sp.heads_start(k, self.num_heads)
            hv = sp.heads_start(linear(xin, self.v), self.num_heads)
            q = linear(xin, self.q)
            ops.rmsnorm_rope(q, self.norm_q.weight, q_cos, q_sin, self.head_dim, self.norm_q.eps)
            hq = sp.heads_start(q, self.num_heads)
            return sp.attention_started(hq, hk, hv, self.num_heads, tuple(q.shape), scale=attn_scale)
        q, k = linear(xin, self.q), linear(xin, self.k)
        if keep is not None and keep.get("wide"):      # training with room to spare: the pre-norm projections stay for the backward
            keep["qp"], keep["kp"] = q, k
            q, k = q.clone(), k.clone()
        ops.rmsnorm_rope(q, self.norm_q.weight, q_cos, q_sin, self.head_dim, self.norm_q.eps)
        ops.rmsnorm_rope(k, self.norm_k.weight, rope.cos, rope.sin, self.head_dim, self.norm_k.eps)
        if sp is None and keep is None and not fp8 and x2.is_cuda and self.v.weight.shape[0] >= 512 \
                and ops.vt32_ok(x2.shape[0], self.num_heads, self.head_dim):
            # inference on one GPU: nothing but the attention reads V, so the projection writes it straight in the layout the
            # attention kernel wants (gf_linear_vt32: same bits as the plain projection + the transpose, one pass over V less)
            return ops.flash_attn(q, k, None, self.num_heads, vt=ops.linear_vt32(x2, self.v.weight, self.v.bias), scale=attn_scale)
        if sp is None and keep is None and fp8 and xin.is_cuda and self.v.weight.shape[0] >= 512 and self.v.weight.shape[1] % 128 == 0 \
                and ops.vt32_ok(xin.shape[0], self.num_heads, self.head_dim):
            # config 5 on one GPU: the fp8 V projection writes the attention kernel's V^T operand as well (gf_linear_vt32_fp8)
            if self.v._gf_w8_key != param_key(self.v.weight):
                self.v._gf_w8 = ops.cast_fp8(self.v.weight.detach().contiguous())
                self.v._gf_w8_key = param_key(self.v.weight)
            return ops.flash_attn(q, k, None, self.num_heads, vt=ops.linear_vt32_fp8(xin.x8, xin.scale, self.v._gf_w8, self.v.bias), scale=attn_scale)
        v = linear(xin, self.v)
        if sp is not None:
            return sp.attention(q, k, v, self.num_heads, scale=attn_scale)
        if keep is not None:
            keep["attn"], keep["lse"] = ops.flash_attn_lse(q, k, v, self.num_heads, scale=attn_scale)
            if keep.get("wide"):
                keep["v"] = v
            return keep["attn"]
        return ops.flash_attn(q, k, v, self.num_heads, scale=attn_scale)

    def forward(self, x, freqs):
        x2 = _tokens2d(x)
        a = self.attend(x2, _as_rope(freqs, x.device))
        return linear(a, self.o).view(x.shape)


class CrossAttention(nn.Module):
    """DIT:150-186, has_image_input=False branch (A14B I2V: DIT:703-718)."""

    def __init__(self, dim: int, num_heads: int, eps: float = 1e-6, has_image_input: bool = False):
        super().__init__()
        if has_image_input:
            raise NotImplementedError("has_image_input=True (CLIP image tokens) is outside the Goal-Force path")
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = RMSNorm(dim, eps=eps), RMSNorm(dim, eps=eps)
        self.has_image_input = False

    def context_kv(self, ctx2: torch.Tensor, fold: bool = True, pad_n: Optional[int] = None):
        """k = norm_k(Wk ctx), v = Wv ctx — constant per (expert, prompt, block); cacheable over steps.  -> (k, v, m).

        `fold`: the prompter zeroes the text-encoder output past the prompt (wan_prompter.py:99-109), so the padded rows of
        `text_embedding(context)` are all the same vector — and so are their K and V rows (every op here acts row by row).
        softmax over [k_0 .. k_{n-1}, k_pad x m] is evaluated on n + 1 keys with the last one counting m times
        (ops.flash_attn(last_key_mult=m)): the same function of q, 41 keys instead of 512 for a 40-token prompt.  The run is
        DETECTED on the tensor (pad_run), never assumed; a context without such a run is attended in full.  `pad_n` is that
        detection's result when the caller already ran it on this very tensor (model_fn does, once per embedded context, instead
        of one host read-back per block).  `ops.options(fold_pad_keys=False)` switches the folding off (A/B, and the cross-check in
        tests/test_kernels_gpu.py)."""
        m = 1
        if fold and ctx2.is_cuda and ops._OPT["fold_pad_keys"]:
            n = pad_run(ctx2) if pad_n is None else pad_n
            if ctx2.shape[0] - n >= 2 and n + 1 < ops.VT_MIN_KV:      # (the multiplicity form exists for short key sequences only)
                m = ctx2.shape[0] - n
                ctx2 = ctx2[: n + 1]
        cin = QuantizedInput(ctx2) if getattr(self.k, "_gf_w8", None) is not None else ctx2
        k, v = linear(cin, self.k), linear(cin, self.v)
        ops.rmsnorm_rope(k, self.norm_k.weight, None, None, self.head_dim, self.norm_k.eps)
        return k, v, m

    def attend(self, x2, kv):
        q = linear(x2, self.q)
        ops.rmsnorm_rope(q, self.norm_q.weight, None, None, self.head_dim, self.norm_q.eps)
        return ops.flash_attn(q, kv[0], kv[1], self.num_heads, last_key_mult=kv[2] if len(kv) > 2 else 1)

    def forward(self, x: torch.Tensor, y: torch.Tensor):
        x2 = _tokens2d(x)
        a = self.attend(x2, self.context_kv(_tokens2d(y)))
        return linear(a, self.o).view(x.shape)


class GateModule(nn.Module):
    """DIT:189-194 — `x + gate * residual`.  DiTBlock.forward fuses this into the epilogue of the GEMM that produces `residual`
    (GF_EPI_BIAS_GATE_RESID); the module's own forward is the stand-alone kernel gf_gate_residual with the same bf16 rounding
    order (the product first, then the sum), for callers that use the module directly."""

    def forward(self, x, gate, residual):
        if x.shape != residual.shape:
            raise GoalForceError("GateModule: x and residual must have the same shape")
        d = x.shape[-1]
        g = gate.reshape(-1, d)
        if x.dim() == 3 and g.shape[0] == x.shape[0] and x.shape[0] > 1:        # one gate row per batch element ([B,1,D])
            return torch.stack([ops.gate_residual(x[b], g[b].contiguous(), residual[b]) for b in range(x.shape[0])])
        if g.shape[0] != 1:
            raise GoalForceError("GateModule: gate must be [B or 1, 1, D] (per-token gates are outside the Goal-Force path)")
        return ops.gate_residual(x, g[0].contiguous(), residual)


class DiTBlock(nn.Module):
    """DIT:197-230.  forward(x, context, t_mod, freqs) -> new x; x [1,S,D] bf16, context [1,L,D],
    t_mod [1,6,D], freqs RopeTable (or the reference's complex tensor)."""

    def __init__(self, has_image_input: bool, dim: int, num_heads: int, ffn_dim: int, eps: float = 1e-6):
        super().__init__()
        self.dim, self.num_heads, self.ffn_dim, self.eps = dim, num_heads, ffn_dim, eps
        self.self_attn = SelfAttention(dim, num_heads, eps)
        self.cross_attn = CrossAttention(dim, num_heads, eps, has_image_input=has_image_input)
        self.norm1 = nn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.norm2 = nn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.norm3 = nn.LayerNorm(dim, eps=eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        self.gate = GateModule()

    def forward(self, x, context, t_mod, freqs, context_kv=None, out=None, sp=None, keep=None, self_attn_memo=None,
                fold_pad_keys=True, pad_n=None):
        """`self_attn_memo` (a dict, optional): x + gate_msa * self_attn(modulate(norm1(x))) — the block's first half — does not
        depend on the text context.  The two forwards of a CFG step run this block on IDENTICAL x, t_mod and freqs (block 0 of the
        DiT and of the ControlNet, model_fn_wan_video), so the first stores that half here and the second takes it: same kernels
        on the same inputs, the same bits, one attention and four projections fewer."""
        if t_mod.dim() == 4:
            raise NotImplementedError("per-token t_mod (seperated_timestep) is outside the Goal-Force path")
        x2 = _tokens2d(x)
        rope = _as_rope(freqs, x.device)
        # rows: shift_msa, 1+scale_msa, gate_msa, shift_mlp, 1+scale_mlp, gate_mlp   (DIT:218-219, 64-65)
        mod = ops.modulation(self.modulation, t_mod.contiguous(), onep_mask=0b010010)
        # config 5: every Linear of the block on the fp8_linear contract — all of them or none (enable_fp8)
        fp8 = getattr(self.ffn[0], "_gf_w8", None) is not None
        if self_attn_memo is not None and "x" in self_attn_memo:
            x_new = self_attn_memo.pop("x")
            ev = self_attn_memo.pop("ev", None)
            if ev is not None:      # the other CFG branch may run on another stream (pipeline.cfg_streams): order after its write
                torch.cuda.current_stream(x_new.device).wait_event(ev)
                x_new.record_stream(torch.cuda.current_stream(x_new.device))
            if out is not None:
                x_new = out.copy_(x_new)
            h = None if fp8 else torch.empty_like(x2)
        else:
            if fp8:                              # the normalised row goes straight to e4m3 + scale_a (never to HBM as bf16)
                h = QuantizedInput.layernorm(x2, scale1p=mod[1], shift=mod[0], eps=self.eps)
            else:
                h = ops.layernorm_modulate(x2, scale1p=mod[1], shift=mod[0], eps=self.eps)          # DIT:225
            a = self.self_attn.attend(h, rope, sp, keep)
            x_new = out if out is not None else torch.empty_like(x2)
            linear(a, self.self_attn.o, epilogue=ops.EPI_BIAS_GATE_RESID, resid=x2, gate=mod[2], out=x_new)   # DIT:226
            if keep is not None and keep.get("wide"):
                keep["x1"] = x_new.clone()               # (the rest of the block updates x_new in place)
            if self_attn_memo is not None:
                self_attn_memo["x"] = x_new.clone()      # the rest of the block updates x_new in place
                if x_new.is_cuda:
                    self_attn_memo["ev"] = torch.cuda.Event()
                    self_attn_memo["ev"].record(torch.cuda.current_stream(x_new.device))
        if fp8:
            h = QuantizedInput.layernorm(x_new, weight=self.norm3.weight, bias=self.norm3.bias, eps=self.eps)
        else:
            ops.layernorm_modulate(x_new, weight=self.norm3.weight, bias=self.norm3.bias, eps=self.eps, out=h)
        if context_kv is None:
            context_kv = self.cross_attn.context_kv(_tokens2d(context), fold=fold_pad_keys and keep is None, pad_n=pad_n)
        a = self.cross_attn.attend(h, context_kv)
        linear(a, self.cross_attn.o, epilogue=ops.EPI_BIAS_RESID, resid=x_new, out=x_new)               # DIT:227
        if keep is not None and keep.get("wide"):
            keep["x2b"] = x_new.clone()
        if fp8:
            h = QuantizedInput.layernorm(x_new, scale1p=mod[4], shift=mod[3], eps=self.eps)
        else:
            ops.layernorm_modulate(x_new, scale1p=mod[4], shift=mod[3], eps=self.eps, out=h)     # DIT:228
        f1 = linear(h, self.ffn[0], epilogue=ops.EPI_BIAS_GELU_TANH)
        linear(f1, self.ffn[2], epilogue=ops.EPI_BIAS_GATE_RESID, resid=x_new, gate=mod[5], out=x_new)  # DIT:229
        return x_new.view(x.shape)


class Head(nn.Module):
    """DIT:253-269."""

    def __init__(self, dim: int, out_dim: int, patch_size: Tuple[int, int, int], eps: float):
        super().__init__()
        self.dim, self.patch_size, self.eps = dim, patch_size, eps
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.head = nn.Linear(dim, out_dim * math.prod(patch_size))
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, t_mod):
        if t_mod.dim() == 3:
            raise NotImplementedError("per-token head modulation is outside the Goal-Force path")
        x2 = _tokens2d(x)
        mod = ops.modulation(self.modulation, t_mod.contiguous(), onep_mask=0b10)   # rows: shift, 1+scale
        h = ops.layernorm_modulate(x2, scale1p=mod[1], shift=mod[0], eps=self.eps)
        y = ops.gemm(h, self.head.weight, self.head.bias)
        return y.view(x.shape[:-1] + (y.shape[-1],))


class WanModel(nn.Module):
    """DIT:272-340.  A14B I2V config: dim 5120, in_dim 36, ffn 13824, out 16, text 4096, freq 256,
    40 heads, 40 layers, has_image_input=False, require_clip_embedding=False (DIT:703-718)."""

    def __init__(self, dim: int, in_dim: int, ffn_dim: int, out_dim: int, text_dim: int, freq_dim: int, eps: float,
                 patch_size: Tuple[int, int, int], num_heads: int, num_layers: int, has_image_input: bool,
                 has_image_pos_emb: bool = False, has_ref_conv: bool = False, add_control_adapter: bool = False,
                 in_dim_control_adapter: int = 24, seperated_timestep: bool = False,
                 require_vae_embedding: bool = True, require_clip_embedding: bool = True,
                 fuse_vae_embedding_in_latents: bool = False):
        super().__init__()
        if has_image_input or has_ref_conv or add_control_adapter or seperated_timestep:
            raise NotImplementedError("only the Wan2.2-I2V-A14B configuration (DIT:703-718) is built")
        if tuple(patch_size) != (1, 2, 2):
            raise NotImplementedError("patch_size must be (1,2,2)")
        self.dim, self.in_dim, self.freq_dim, self.out_dim = dim, in_dim, freq_dim, out_dim
        self.has_image_input = has_image_input
        self.patch_size = tuple(patch_size)
        self.seperated_timestep = seperated_timestep
        self.require_vae_embedding = require_vae_embedding
        self.require_clip_embedding = require_clip_embedding
        self.fuse_vae_embedding_in_latents = fuse_vae_embedding_in_latents
        self.num_heads = num_heads
        self.eps = eps
        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=patch_size, stride=patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList([DiTBlock(has_image_input, dim, num_heads, ffn_dim, eps) for _ in range(num_layers)])
        self.head = Head(dim, out_dim, patch_size, eps)
        self.freqs = precompute_freqs_cis_3d(dim // num_heads)  # plain CPU tuple, like DIT:328
        self.control_adapter = None
        self._rope_cache = {}
        self._patch_w, self._patch_w_key = None, None  # K-padded [dim, kpad] copy of patch_embedding.weight for the GEMM

    # ---- pieces used by model_fn ---------------------------------------------------------------
    def rope_table(self, f, h, w, device) -> RopeTable:
        key = (f, h, w, str(device))
        if key not in self._rope_cache:
            self._rope_cache[key] = RopeTable.from_grid(self.freqs, f, h, w, device)
        return self._rope_cache[key]

    def rope_table_shard(self, f, h, w, device, sp) -> RopeTable:
        """This rank's slice of the table under sequence parallelism (cached like the full one)."""
        key = (f, h, w, str(device), "sp", sp.rank, sp.size)
        if key not in self._rope_cache:
            self._rope_cache[key] = sp.shard_rope(self.rope_table(f, h, w, device))
        return self._rope_cache[key]

    def time_embed(self, timestep: torch.Tensor):
        """GF:1441-1442 — t [1,D], t_mod [1,6,D]."""
        e = sinusoidal_embedding_1d(self.freq_dim, timestep)
        te = self.time_embedding
        t = ops.gemm(ops.gemm(e, te[0].weight, te[0].bias, epilogue=ops.EPI_BIAS_SILU), te[2].weight, te[2].bias)
        tp = self.time_projection[1]
        t_mod = ops.gemm(ops.act(t, "silu"), tp.weight, tp.bias).unflatten(1, (6, self.dim))
        return t, t_mod

    def embed_text(self, context: torch.Tensor):
        """DIT:309-313 — [1,L,text_dim] -> [1,L,D]."""
        te = self.text_embedding
        c2 = _tokens2d(context)
        y = ops.gemm(ops.gemm(c2, te[0].weight, te[0].bias, epilogue=ops.EPI_BIAS_GELU_TANH), te[2].weight, te[2].bias)
        return y.view(context.shape[:-1] + (self.dim,))

    @staticmethod
    def padded_patch_weight(conv: nn.Conv3d) -> torch.Tensor:
        w = conv.weight
        k = w.shape[1] * 4
        kpad = -(-k // 64) * 64
        wp = torch.zeros((w.shape[0], kpad), dtype=w.dtype, device=w.device)
        wp[:, :k] = w.reshape(w.shape[0], k)
        return wp

    def patchify(self, x: torch.Tensor, control_camera_latents_input: Optional[torch.Tensor] = None, extra=None):
        """DIT:341-349 — Conv3d k=s=(1,2,2) as im2col gather + MFMA GEMM; returns ([1,S,D], (f,h,w)).
        `extra` lets model_fn pass y without materialising cat([latents, y]) (GF:1458)."""
        if control_camera_latents_input is not None:
            raise NotImplementedError("camera control adapter is outside the Goal-Force path")
        if x.shape[0] != 1:
            raise GoalForceError("patchify takes batch 1")
        if self._patch_w is None or self._patch_w_key != param_key(self.patch_embedding.weight):
            self._patch_w = self.padded_patch_weight(self.patch_embedding)
            self._patch_w_key = param_key(self.patch_embedding.weight)
        s0 = x[0].contiguous()
        s1 = None if extra is None else extra[0].contiguous()
        f, hh, ww = s0.shape[1], s0.shape[2] // 2, s0.shape[3] // 2
        cols = ops.patchify_im2col(s0, s1, kpad=self._patch_w.shape[1])
        tok = ops.gemm(cols, self._patch_w, self.patch_embedding.bias)
        return tok.unsqueeze(0), (f, hh, ww)

    def unpatchify(self, x: torch.Tensor, grid_size):
        """DIT:351-356."""
        f, h, w = grid_size
        return ops.unpatchify(_tokens2d(x).contiguous(), self.out_dim, f, h, w).unsqueeze(0)

    def _load_from_state_dict(self, *args, **kwargs):
        self._patch_w = None
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, timestep, context, clip_feature=None, y=None, **kwargs):
        """DIT:358-... plain forward without ControlNet (model_fn is the Goal-Force entry point)."""
        from .model_fn import model_fn_wan_video
        return model_fn_wan_video(self, latents=x, timestep=timestep, context=context, clip_feature=clip_feature, y=y)

    @staticmethod
    def state_dict_converter():
        raise NotImplementedError("checkpoint key conversion is host-side I/O outside the hot path (SURVEY §2 #9)")


A14B_CONFIG = dict(has_image_input=False, patch_size=(1, 2, 2), in_dim=36, dim=5120, ffn_dim=13824, freq_dim=256,
                   text_dim=4096, out_dim=16, num_heads=40, num_layers=40, eps=1e-6,
                   require_clip_embedding=False)  # DIT:703-718
