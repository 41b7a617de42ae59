"""Torch-tensor front end of the C ABI (include/goalforce.h).

Every function enqueues exactly the named HIP kernel on torch's current stream; tensors
are only used for device memory + stream plumbing (data_ptr()), never for the arithmetic.
There is no CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_BIAS_GATE_RESID, EPI_BIAS_GELU_TANH, EPI_BIAS_MUL, EPI_BIAS_RESID, EPI_BIAS_SILU,
                   GoalForceError)

__all__ = [
    "modulation", "layernorm_modulate", "rmsnorm_rope", "gate_residual", "gemm", "flash_attn", "patchify_im2col", "unpatchify",
    "cfg_euler_step", "act", "add", "force_map",
    "EPI_BIAS", "EPI_BIAS_GELU_TANH", "EPI_BIAS_GATE_RESID", "EPI_BIAS_RESID", "EPI_BIAS_SILU", "EPI_BIAS_MUL",
]

_BF16 = torch.bfloat16

# Optional per-launch timing of the dominant kernel (bench.py's roofline leg): when PROFILE_ATTN is a list,
# flash_attn() brackets every launch with events recorded on the launch stream and appends
# (start, end, q_len, kv_len, heads).  None (default) = no events, no overhead.
PROFILE_ATTN = None
# Same for the GEMM (bench.py's second roofline entry): a list collects (start, end, M, N, K, epilogue) per launch.
PROFILE_GEMM = None
# the same for the VAE convolutions (bench.py's `roofline_vae`): (ev0, ev1, rows, N, C, kt, ks, mode, with_resid) per gf_conv3d_bf16 launch
PROFILE_CONV = None


def _stream(t: torch.Tensor):
    """The HIP stream the launch goes to: torch's current stream of the tensor's device.  The C ABI launches on the
    CURRENT HIP device, so a tensor of another device is refused instead of being launched on the wrong GPU."""
    if t.device.index != torch.cuda.current_device():
        raise GoalForceError(f"tensor lives on cuda:{t.device.index} but the current device is cuda:"
                             f"{torch.cuda.current_device()}: call torch.cuda.set_device (one process per GPU)")
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, name: str, dtype=_BF16):
    if not t.is_cuda:
        raise GoalForceError(f"{name}: tensor must be on the GPU (no CPU fallback exists)")
    if t.dtype != dtype:
        raise GoalForceError(f"{name}: expected dtype {dtype}, got {t.dtype}")


def _rows2d(t: torch.Tensor, name: str):
    """Return (tensor, rows, dim, row_stride) for a tensor whose last dim is contiguous and whose
    leading dims collapse to one row index with a single stride."""
    if t.stride(-1) != 1:
        raise GoalForceError(f"{name}: last dim must be contiguous")
    if t.dim() == 1:
        return t, 1, t.shape[0], t.shape[0]
    v = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t
    if v.dim() != 2:
        raise GoalForceError(f"{name}: cannot view as [rows, dim] without a copy")
    return v, v.shape[0], v.shape[1], v.stride(0)


def modulation(param, t, onep_mask: int):
    """out[i] = bf16(param[i] + t[i % t_rows]); rows in onep_mask get bf16(1 + .) — gf_modulation.
    param [.., k, dim], t [.., t_rows, dim] -> out [k, dim]."""
    _req(param, "modulation.param")
    _req(t, "modulation.t")
    dim = param.shape[-1]
    k = param.numel() // dim
    t_rows = t.numel() // dim
    if not param.is_contiguous() or not t.is_contiguous() or t.shape[-1] != dim:
        raise GoalForceError("modulation: param/t must be contiguous with equal last dim")
    out = torch.empty((k, dim), dtype=_BF16, device=param.device)
    _lib.check(_lib.load().gf_modulation(_ptr(param), _ptr(t), _ptr(out), k, dim, t_rows, int(onep_mask),
                                         _stream(param)), "gf_modulation")
    return out


def layernorm_modulate(x, weight=None, bias=None, scale1p=None, shift=None, eps=1e-6, out=None):
    """LayerNorm(+affine)(+modulate) — gf_layernorm_modulate."""
    _req(x, "layernorm_modulate.x")
    xv, rows, dim, xs = _rows2d(x, "layernorm_modulate.x")
    if out is None:
        out = torch.empty(x.shape, dtype=_BF16, device=x.device)
    ov, _, _, os_ = _rows2d(out, "layernorm_modulate.out")
    for n, t in (("weight", weight), ("bias", bias), ("scale1p", scale1p), ("shift", shift)):
        if t is not None:
            _req(t, f"layernorm_modulate.{n}")
            if t.numel() != dim or not t.is_contiguous():
                raise GoalForceError(f"layernorm_modulate.{n}: expected contiguous [{dim}]")
    _lib.check(_lib.load().gf_layernorm_modulate(_ptr(xv), _ptr(ov), _ptr(weight), _ptr(bias), _ptr(scale1p),
                                                 _ptr(shift), rows, dim, xs, os_, float(eps), _stream(x)),
               "gf_layernorm_modulate")
    return out


def rmsnorm_rope(x, weight, cos=None, sin=None, head_dim=128, eps=1e-6):
    """In-place full-width RMSNorm (+RoPE) — gf_rmsnorm_rope."""
    _req(x, "rmsnorm_rope.x")
    _req(weight, "rmsnorm_rope.weight")
    xv, rows, dim, xs = _rows2d(x, "rmsnorm_rope.x")
    if cos is not None:
        _req(cos, "rmsnorm_rope.cos", torch.float32)
        _req(sin, "rmsnorm_rope.sin", torch.float32)
        if cos.shape != (rows, head_dim // 2) or sin.shape != cos.shape or not cos.is_contiguous() \
                or not sin.is_contiguous():
            raise GoalForceError(f"rmsnorm_rope: cos/sin must be contiguous [{rows}, {head_dim // 2}]")
    _lib.check(_lib.load().gf_rmsnorm_rope(_ptr(xv), _ptr(weight), _ptr(cos), _ptr(sin), rows, dim, head_dim, xs,
                                           float(eps), _stream(x)), "gf_rmsnorm_rope")
    return x


def modulate(x, shift, scale):
    """x * (1 + scale) + shift in the reference's eager bf16 rounding sequence (DIT:64-65) — gf_modulate; shift / scale [dim]."""
    _req(x, "modulate.x")
    xv, rows, dim, xs = _rows2d(x, "modulate.x")
    for n, t in (("shift", shift), ("scale", scale)):
        _req(t, f"modulate.{n}")
        if t.numel() != dim or not t.is_contiguous():
            raise GoalForceError(f"modulate.{n}: expected contiguous [{dim}] (batch 1)")
    out = torch.empty(x.shape, dtype=_BF16, device=x.device)
    ov, _, _, os_ = _rows2d(out, "modulate.out")
    _lib.check(_lib.load().gf_modulate(_ptr(xv), _ptr(ov), _ptr(scale), _ptr(shift), rows, dim, xs, os_, _stream(x)), "gf_modulate")
    return out


def gate_residual(x, gate, residual):
    """x + gate * residual in the reference's eager bf16 rounding order (DIT:189-194) — gf_gate_residual; gate [dim]."""
    _req(x, "gate_residual.x")
    _req(residual, "gate_residual.residual")
    _req(gate, "gate_residual.gate")
    xv, rows, dim, xs = _rows2d(x, "gate_residual.x")
    rv, rrows, rdim, rs = _rows2d(residual, "gate_residual.residual")
    if (rrows, rdim) != (rows, dim) or gate.numel() != dim or not gate.is_contiguous():
        raise GoalForceError(f"gate_residual: residual must match x [{rows}, {dim}] and gate be contiguous [{dim}] (batch 1)")
    out = torch.empty(x.shape, dtype=_BF16, device=x.device)
    ov, _, _, os_ = _rows2d(out, "gate_residual.out")
    _lib.check(_lib.load().gf_gate_residual(_ptr(xv), _ptr(gate), _ptr(rv), _ptr(ov), rows, dim, xs, rs, os_, _stream(x)),
               "gf_gate_residual")
    return out


def rope_apply(x, cos, sin, head_dim):
    """Rotary embedding alone (DIT:92-97) — gf_rope_apply; x [S, H*head_dim] bf16, cos / sin [S, head_dim/2] fp32 -> new tensor."""
    _req(x, "rope_apply.x")
    xv, rows, dim, xs = _rows2d(x, "rope_apply.x")
    for n, t in (("cos", cos), ("sin", sin)):
        _req(t, f"rope_apply.{n}", torch.float32)
        if tuple(t.shape) != (rows, head_dim // 2) or not t.is_contiguous():
            raise GoalForceError(f"rope_apply.{n}: expected contiguous [{rows}, {head_dim // 2}] fp32")
    out = torch.empty(x.shape, dtype=_BF16, device=x.device)
    ov, _, _, os_ = _rows2d(out, "rope_apply.out")
    _lib.check(_lib.load().gf_rope_apply(_ptr(xv), _ptr(ov), _ptr(cos), _ptr(sin), rows, dim, head_dim, xs, os_, _stream(x)),
               "gf_rope_apply")
    return out


def gemm(a, w, bias=None, epilogue=EPI_BIAS, resid=None, gate=None, out=None):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T + bias) — gf_gemm_bf16."""
    _req(a, "gemm.a")
    _req(w, "gemm.w")
    av, M, K, lda = _rows2d(a, "gemm.a")
    if w.dim() != 2 or w.stride(1) != 1 or w.shape[1] != K:
        raise GoalForceError(f"gemm.w: expected [N, {K}] with contiguous rows, got {tuple(w.shape)}")
    N = w.shape[0]
    if out is None:
        out = torch.empty(a.shape[:-1] + (N,), dtype=_BF16, device=a.device)
    ov, Mo, No, ldc = _rows2d(out, "gemm.out")
    if (Mo, No) != (M, N):
        raise GoalForceError(f"gemm.out: expected [{M}, {N}], got [{Mo}, {No}]")
    ldr = 0
    if resid is not None:
        _req(resid, "gemm.resid")
        rv, Mr, Nr, ldr = _rows2d(resid, "gemm.resid")
        if (Mr, Nr) != (M, N):
            raise GoalForceError("gemm.resid: shape mismatch")
        resid = rv
    if gate is not None:
        _req(gate, "gemm.gate")
        if gate.numel() != N or not gate.is_contiguous():
            raise GoalForceError(f"gemm.gate: expected contiguous [{N}]")
    if bias is not None:
        _req(bias, "gemm.bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise GoalForceError(f"gemm.bias: expected contiguous [{N}]")
    prof = PROFILE_GEMM
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().gf_gemm_bf16(_ptr(av), lda, _ptr(w), w.stride(0), _ptr(bias), _ptr(ov), ldc, M, N, K,
                                        int(epilogue), _ptr(resid), ldr, _ptr(gate), _stream(a)), "gf_gemm_bf16")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, M, N, K, int(epilogue)))
    return out


VT_MIN_KV = 2048   # key sequences at least this long go through the pre-transposed-V kernel (kernel 3)
# Dispatch overrides.  Every kernel ships, each for the shapes its launcher sends it; the parity tests cross-check two kernels on
# the SAME operands, which needs a way to route a shape to the kernel that would not get it by default.  Nothing here (and nothing
# in the library) reads an environment variable: the defaults are the shipped dispatch and only `options(...)` changes them.
#   library side (gf_set_option): prefer_8wave, a4_stagger, a4_group_m, conv_nb, conv_gather, conv_direct, vae_rms3
#   this module: attn_k3 (False: long key sequences on kernel 2), vt_from_gemm (False: V^T by gf_transpose_v32, not by the V projection),
#                conv_padded (False: the VAE's 192 / 384-channel 3x3x3 convolutions on gf_conv3d_bf16 instead of the padded-layout kernel),
#                fold_pad_keys (False: cross-attention over all 512 context keys instead of the prompt + ONE key of multiplicity, dit.py),
#                attn_q_prescale (False: the self-attention's Q is rotated plainly and scaled inside the attention kernel — a second bf16 rounding),
#                vae_attn_offset (False: the VAE attention's softmax on the bf16-rounded RAW scores — one GEMM instead of two, coarser)
_LIB_DEFAULTS = {"prefer_8wave": 0, "a4_stagger": 2, "a4_group_m": 0, "conv_nb": 0, "conv_gather": 0, "conv_direct": 1, "vae_rms3": 1}
_OPT = {"attn_k3": True, "vt_from_gemm": True, "conv_padded": True, "fold_pad_keys": True, "attn_q_prescale": True, "vae_attn_offset": True}


def lib_option(name: str) -> int:
    """The value of a library-side dispatch option as the LIBRARY holds it (gf_get_option) — not a Python mirror: a caller that
    used gf_set_option / gf_reset_options directly is seen."""
    import ctypes
    v = ctypes.c_int(0)
    _lib.check(_lib.load().gf_get_option(name.encode(), ctypes.byref(v)), f"gf_get_option({name})")
    return int(v.value)


class options:
    """`with ops.options(prefer_8wave=1): ...` — dispatch overrides for a block, restored afterwards (tests and A/B tools).  The
    restore values are read from the library when the block is entered; an override that fails half-way rolls back the ones
    already applied before the error propagates."""

    def __init__(self, **kv):
        for k in kv:
            if k not in _OPT and k not in _LIB_DEFAULTS:
                raise GoalForceError(f"ops.options: unknown option {k!r}")
        self.kv, self.old = kv, {}

    @staticmethod
    def _apply_one(k, v):
        if k in _OPT:
            _OPT[k] = bool(v)
        else:
            _lib.check(_lib.load().gf_set_option(k.encode(), int(v)), f"gf_set_option({k})")

    def __enter__(self):
        self.old = {}
        try:
            for k, v in self.kv.items():
                old = _OPT[k] if k in _OPT else lib_option(k)
                self._apply_one(k, v)
                self.old[k] = old
        except BaseException:
            self.__exit__(None, None, None)
            raise
        return self

    def __exit__(self, *exc):
        for k, v in reversed(list(self.old.items())):
            self._apply_one(k, v)
        self.old = {}
        return False


_VT_WS = {}


def _vt_workspace(numel, device):
    """One reusable V^T buffer per (device, stream): on one stream the reuse is ordered (every attention launch that reads
    it is enqueued before the next transpose that overwrites it); launches on another stream get their own buffer."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _VT_WS.get(key)
    if ws is None or ws.numel() < numel:
        ws = _VT_WS[key] = torch.empty((numel,), dtype=_BF16, device=device)
    return ws


def vt32_ok(skv, num_heads, head_dim):
    """True when flash_attn would take the pre-transposed-V kernel 3 for this key length (then linear_vt32 can produce its V^T)."""
    kv_pad = -(-skv // 64) * 64
    return (skv >= VT_MIN_KV and head_dim == 128 and num_heads * 128 * kv_pad < 2 ** 31
            and _OPT["attn_k3"] and _OPT["vt_from_gemm"])


def linear_vt32(x, weight, bias):
    """The V projection of a self-attention, written directly as kernel 3's V^T operand (gf_linear_vt32): returns the V^T
    workspace of this (device, stream) — valid until the next call — for flash_attn(..., vt=...).  Same bits as
    gemm(x, weight, bias) followed by the transpose inside flash_attn."""
    _req(x, "linear_vt32.x")
    _req(weight, "linear_vt32.weight")
    xv, skv, K, ldx = _rows2d(x, "linear_vt32.x")
    if weight.dim() != 2 or weight.stride(1) != 1 or weight.shape[1] != K:
        raise GoalForceError(f"linear_vt32.weight: expected [N, {K}] with contiguous rows")
    N = weight.shape[0]
    if bias is not None:
        _req(bias, "linear_vt32.bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise GoalForceError(f"linear_vt32.bias: expected contiguous [{N}]")
    kv_pad = -(-skv // 64) * 64
    vt = _vt_workspace(N * kv_pad, x.device)
    _lib.check(_lib.load().gf_linear_vt32(_ptr(xv), ldx, _ptr(weight), weight.stride(0), _ptr(bias), _ptr(vt), skv, kv_pad, N, K,
                                          _stream(x)), "gf_linear_vt32")
    return vt


def linear_vt32_fp8(x8, x_scale, w8, bias):
    """linear_vt32 on the fp8_linear contract (config 5): x8 [S, K] float8_e4m3fn + its row scales, w8 [N, K] float8_e4m3fn ->
    the V^T workspace (gf_linear_vt32_fp8); same bits as gemm_fp8(x8, x_scale, w8, bias) followed by the transpose."""
    _req(x8, "linear_vt32_fp8.x8", _FP8)
    _req(w8, "linear_vt32_fp8.w8", _FP8)
    _req(x_scale, "linear_vt32_fp8.x_scale", torch.float32)
    if x8.dim() != 2 or w8.dim() != 2 or x8.stride(1) != 1 or w8.stride(1) != 1 or x8.shape[1] != w8.shape[1]:
        raise GoalForceError("linear_vt32_fp8: x8 [S,K], w8 [N,K] with contiguous rows expected")
    skv, K = x8.shape
    N = w8.shape[0]
    if x_scale.numel() != skv or not x_scale.is_contiguous():
        raise GoalForceError("linear_vt32_fp8.x_scale: contiguous [S] expected")
    if bias is not None:
        _req(bias, "linear_vt32_fp8.bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise GoalForceError(f"linear_vt32_fp8.bias: expected contiguous [{N}]")
    kv_pad = -(-skv // 64) * 64
    vt = _vt_workspace(N * kv_pad, x8.device)
    _lib.check(_lib.load().gf_linear_vt32_fp8(_ptr(x8), x8.stride(0), _ptr(x_scale), _ptr(w8), w8.stride(0), _ptr(bias), _ptr(vt), skv,
                                              kv_pad, N, K, _stream(x8)), "gf_linear_vt32_fp8")
    return vt


def flash_attn(q, k, v, num_heads, out=None, scale=None, vt=None, last_key_mult=1, finished_q=False):
    """softmax(q k^T / sqrt(d)) v per head; q [Sq, H*128], k/v [Skv, H*128] (row-strided views OK).  `vt` (instead of v): the
    V^T operand linear_vt32 produced for these keys.  `last_key_mult` = m > 1: the last key counts m times (a run of m identical
    trailing keys folded into one: gf_flash_attn_fwd_lastmult; key lengths below the V^T threshold only).
    `finished_q`: q is somebody else's final bf16 tensor (the B3 drop-in dit.flash_attention).  Kernel 3 multiplies Q by
    scale x log2(e) and rounds it to bf16 again — 3.9e-3 / 5.2e-3 from fp64 at logit std 3 / 8 where torch's SDPA is at 1.8e-3 /
    1.7e-3; the package's own modules avoid that by producing q pre-scaled (dit.SelfAttention.attend), a caller with a finished q
    cannot: long key sequences then run on kernel 2, which scales the fp32 scores (2.1e-3 / 1.9e-3; +16 % time at S = 32760,
    profiles/r06/b3_precision.log)."""
    for n, t in (("q", q), ("k", k)) + ((("v", v),) if vt is None else ()):
        _req(t, f"flash_attn.{n}")
        if t.dim() != 2 or t.stride(1) != 1:
            raise GoalForceError(f"flash_attn.{n}: expected 2-D [len, heads*head_dim] with contiguous rows")
    sq, hd_all = q.shape
    skv = k.shape[0]
    head_dim = hd_all // num_heads
    if k.shape[1] != hd_all or (vt is None and v.shape != k.shape) or head_dim * num_heads != hd_all:
        raise GoalForceError("flash_attn: q/k/v shape mismatch")
    if vt is not None and not vt32_ok(skv, num_heads, head_dim):
        raise GoalForceError("flash_attn: a pre-transposed V is only taken by kernel 3 (key length >= VT_MIN_KV, head_dim 128)")
    if out is None:
        out = torch.empty((sq, hd_all), dtype=_BF16, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    prof = PROFILE_ATTN
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib = _lib.load()
    kv_pad = -(-skv // 64) * 64
    if last_key_mult != 1 and (vt is not None or skv >= VT_MIN_KV):
        raise GoalForceError("flash_attn: last_key_mult is for short key sequences (the text context), not the V^T path")
    if skv >= VT_MIN_KV and head_dim == 128 and num_heads * 128 * kv_pad < 2 ** 31:
        # long key sequences (the DiT self-attention): hand V over pre-transposed — one LDS read per PV MFMA instead of two;
        # the transpose (0.7 % of the attention's time at S=32760) is inside the timed region
        # kernel 3 (16x16x32 MFMAs); options(attn_k3=False) sends the shape to kernel 2 (32x32x16) for the cross-check tests
        if finished_q and vt is not None:
            raise GoalForceError("flash_attn: finished_q goes to kernel 2, which does not take kernel 3's V^T operand")
        k3 = _OPT["attn_k3"] and not finished_q
        tr, fa = (lib.gf_transpose_v32, lib.gf_flash_attn_fwd_vt32) if k3 else (lib.gf_transpose_v, lib.gf_flash_attn_fwd_vt)
        if vt is None:
            vt = _vt_workspace(num_heads * 128 * kv_pad, q.device)
            _lib.check(tr(_ptr(v), v.stride(0), _ptr(vt), skv, kv_pad, num_heads, _stream(q)), "gf_transpose_v")
        elif vt.numel() < num_heads * 128 * kv_pad:
            raise GoalForceError("flash_attn.vt: buffer smaller than heads*128*kv_pad")
        _lib.check(fa(_ptr(q), _ptr(k), _ptr(vt), _ptr(out), None, sq, skv, kv_pad, num_heads, head_dim,
                      q.stride(0), k.stride(0), out.stride(0), float(scale), _stream(q)), "gf_flash_attn_fwd_vt")
    elif last_key_mult != 1:
        _lib.check(lib.gf_flash_attn_fwd_lastmult(_ptr(q), _ptr(k), _ptr(v), _ptr(out), sq, skv, num_heads, head_dim, q.stride(0),
                                                  k.stride(0), v.stride(0), out.stride(0), float(scale), float(last_key_mult), _stream(q)),
                   "gf_flash_attn_fwd_lastmult")
    else:
        _lib.check(lib.gf_flash_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), sq, skv, num_heads, head_dim,
                                         q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(scale),
                                         _stream(q)), "gf_flash_attn_fwd")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, sq, skv, num_heads))
    return out


def flash_attn_lse(q, k, v, num_heads, scale=None):
    """flash_attn that also returns the log2-domain log-sum-exp [Sq, H] fp32 (what flash_attn_bwd rebuilds P from)."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _req(t, f"flash_attn_lse.{n}")
        if t.dim() != 2 or t.stride(1) != 1:
            raise GoalForceError(f"flash_attn_lse.{n}: expected 2-D [len, heads*head_dim] with contiguous rows")
    sq, hd_all = q.shape
    skv = k.shape[0]
    head_dim = hd_all // num_heads
    if k.shape[1] != hd_all or v.shape != k.shape or head_dim * num_heads != hd_all:
        raise GoalForceError("flash_attn_lse: q/k/v shape mismatch")
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    out = torch.empty((sq, hd_all), dtype=_BF16, device=q.device)
    lse = torch.empty((sq, num_heads), dtype=torch.float32, device=q.device)
    lib = _lib.load()
    kv_pad = -(-skv // 64) * 64
    if skv >= VT_MIN_KV and head_dim == 128 and num_heads * 128 * kv_pad < 2 ** 31:   # as flash_attn: pre-transposed V, same bits
        vt = _vt_workspace(num_heads * 128 * kv_pad, q.device)
        k3 = _OPT["attn_k3"]
        tr, fa = (lib.gf_transpose_v32, lib.gf_flash_attn_fwd_vt32) if k3 else (lib.gf_transpose_v, lib.gf_flash_attn_fwd_vt)
        _lib.check(tr(_ptr(v), v.stride(0), _ptr(vt), skv, kv_pad, num_heads, _stream(q)), "gf_transpose_v")
        _lib.check(fa(_ptr(q), _ptr(k), _ptr(vt), _ptr(out), _ptr(lse), sq, skv, kv_pad, num_heads, head_dim,
                      q.stride(0), k.stride(0), out.stride(0), float(scale), _stream(q)), "gf_flash_attn_fwd_vt")
        return out, lse
    _lib.check(lib.gf_flash_attn_fwd_lse(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), sq, skv, num_heads, head_dim,
                                         q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(scale),
                                         _stream(q)), "gf_flash_attn_fwd_lse")
    return out, lse


def flash_attn_bwd(q, k, v, o, dout, lse, num_heads, scale=None, need_dkv=True):
    """Gradients of flash_attn: (dq [Sq, H*128], dk, dv [Skv, H*128]) bf16; need_dkv=False: (dq, None, None) — the dK/dV kernel is
    not launched (a frozen block's cross-attention: nobody reads the gradients of the text context)."""
    for n, t in (("q", q), ("k", k), ("v", v), ("o", o), ("dout", dout)):
        _req(t, f"flash_attn_bwd.{n}")
        if t.dim() != 2 or t.stride(1) != 1:
            raise GoalForceError(f"flash_attn_bwd.{n}: expected 2-D [len, heads*head_dim] with contiguous rows")
    sq, hd_all = q.shape
    skv = k.shape[0]
    head_dim = hd_all // num_heads
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    if lse.dtype != torch.float32 or tuple(lse.shape) != (sq, num_heads) or not lse.is_contiguous():
        raise GoalForceError("flash_attn_bwd.lse: expected contiguous fp32 [Sq, heads]")
    dq = torch.empty_like(q, memory_format=torch.contiguous_format)
    dk = torch.empty((skv, hd_all), dtype=_BF16, device=q.device) if need_dkv else None
    dv = torch.empty((skv, hd_all), dtype=_BF16, device=q.device) if need_dkv else None
    lib = _lib.load()
    # rowsum(dout * o) and the transposed copies of k, q, dout the kernels stream; scratch, sized by the library
    ws = torch.empty((int(lib.gf_flash_attn_bwd_workspace_bytes(sq, skv, num_heads)),), dtype=torch.uint8, device=q.device)
    _lib.check(lib.gf_flash_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(dout), _ptr(lse), _ptr(ws), _ptr(dq),
                                             _ptr(dk) if need_dkv else None, _ptr(dv) if need_dkv else None, sq, skv, num_heads, head_dim,
                                             q.stride(0), k.stride(0), v.stride(0), o.stride(0), dout.stride(0), dq.stride(0),
                                             dk.stride(0) if need_dkv else hd_all, dv.stride(0) if need_dkv else hd_all, float(scale),
                                             _stream(q)), "gf_flash_attn_bwd")
    return dq, dk, dv


def patchify_im2col(src0, src1=None, kpad=None):
    """[c,F,H,W] (+[c1,F,H,W]) -> [F*(H/2)*(W/2), kpad] token-major patches (Conv3d weight column order)."""
    _req(src0, "patchify.src0")
    if src0.dim() != 4 or not src0.is_contiguous():
        raise GoalForceError("patchify.src0: expected contiguous [c,F,H,W]")
    c0, F, H, W = src0.shape
    c1 = 0
    if src1 is not None:
        _req(src1, "patchify.src1")
        if src1.dim() != 4 or not src1.is_contiguous() or src1.shape[1:] != src0.shape[1:]:
            raise GoalForceError("patchify.src1: expected contiguous [c1,F,H,W] matching src0")
        c1 = src1.shape[0]
    if kpad is None:
        kpad = -(-((c0 + c1) * 4) // 64) * 64
    out = torch.empty((F * (H // 2) * (W // 2), kpad), dtype=_BF16, device=src0.device)
    _lib.check(_lib.load().gf_patchify_im2col(_ptr(src0), c0, _ptr(src1), c1, _ptr(out), F, H, W, kpad,
                                              _stream(src0)), "gf_patchify_im2col")
    return out


def unpatchify(tokens, c, f, h, w):
    """[f*h*w, 4c] -> [c, f, 2h, 2w]."""
    _req(tokens, "unpatchify.tokens")
    if tokens.shape != (f * h * w, 4 * c) or not tokens.is_contiguous():
        raise GoalForceError(f"unpatchify.tokens: expected contiguous [{f * h * w}, {4 * c}]")
    out = torch.empty((c, f, 2 * h, 2 * w), dtype=_BF16, device=tokens.device)
    _lib.check(_lib.load().gf_unpatchify(_ptr(tokens), _ptr(out), c, f, h, w, _stream(tokens)), "gf_unpatchify")
    return out


def cfg_euler_step(latents, posi, nega, cfg_scale, dsigma):
    """In place: latents += (nega + cfg*(posi-nega)) * dsigma with bf16 rounding after every op."""
    _req(latents, "cfg_euler_step.latents")
    _req(posi, "cfg_euler_step.posi")
    if not latents.is_contiguous() or not posi.is_contiguous() or posi.numel() != latents.numel():
        raise GoalForceError("cfg_euler_step: latents/posi must be contiguous and equal-sized")
    if nega is not None:
        _req(nega, "cfg_euler_step.nega")
        if not nega.is_contiguous() or nega.numel() != latents.numel():
            raise GoalForceError("cfg_euler_step: nega must be contiguous and equal-sized")
    _lib.check(_lib.load().gf_cfg_euler_step(_ptr(latents), _ptr(posi), _ptr(nega), float(cfg_scale), float(dsigma),
                                             latents.numel(), _stream(latents)), "gf_cfg_euler_step")
    return latents


def act(x, kind: str):
    _req(x, "act.x")
    if not x.is_contiguous():
        raise GoalForceError("act.x must be contiguous")
    out = torch.empty_like(x)
    _lib.check(_lib.load().gf_act(_ptr(x), _ptr(out), x.numel(), {"silu": 0, "gelu_tanh": 1}[kind], _stream(x)),
               "gf_act")
    return out


def add(a, b, out=None):
    _req(a, "add.a")
    _req(b, "add.b")
    if not a.is_contiguous() or not b.is_contiguous() or a.numel() != b.numel():
        raise GoalForceError("add: a/b must be contiguous and equal-sized")
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.load().gf_add_bf16(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream(a)), "gf_add_bf16")
    return out


def force_map(frames, H, W, channels, params, centers, clamp01, device):
    """Render the control-signal video [frames,H,W,3] bf16; blob arrays are torch tensors on `device`."""
    out = torch.empty((frames, H, W, 3), dtype=_BF16, device=device)
    n = int(channels.numel())
    if n:
        _req(channels, "force_map.channels", torch.int32)
        _req(params, "force_map.params", torch.float32)
        _req(centers, "force_map.centers", torch.float32)
        if params.shape != (n, 2) or centers.shape != (n, frames, 2) or not params.is_contiguous() \
                or not centers.is_contiguous():
            raise GoalForceError("force_map: params [n,2] / centers [n,frames,2] expected")
    _lib.check(_lib.load().gf_force_map(_ptr(out), frames, H, W, _ptr(channels) if n else None,
                                        _ptr(params) if n else None, _ptr(centers) if n else None, n,
                                        1 if clamp01 else 0, _stream(out)), "gf_force_map")
    return out


# ---------------------------------------------------------------------------------- VAE decoder kernels
def vae_prep_latent(z_slice, mean, inv_std, cpad=64):
    """z_slice: [C,T,H,W] bf16 view (any strides) -> channels-last [T,H,W,cpad], un-normalised."""
    _req(z_slice, "vae_prep_latent.z")
    _req(mean, "vae_prep_latent.mean")
    _req(inv_std, "vae_prep_latent.inv_std")
    C, T, H, W = z_slice.shape
    out = torch.empty((T, H, W, cpad), dtype=_BF16, device=z_slice.device)
    sc, st, sy, sx = z_slice.stride()
    _lib.check(_lib.load().gf_vae_prep_latent(_ptr(z_slice), sc, st, sy, sx, _ptr(mean), _ptr(inv_std), _ptr(out), C, T,
                                              H, W, cpad, _stream(z_slice)), "gf_vae_prep_latent")
    return out


def vae_im2col(src, cache, kt, ks, kpad, upsample2x=False, downsample2=False, t_stride=1, t_off=0, t_out=None):
    """src [T,H,W,C] (+cache [2,H,W,C]) -> [T_out*Ho*Wo, kpad] patch matrix (modes: see gf_vae_im2col)."""
    _req(src, "vae_im2col.src")
    if not src.is_contiguous():
        raise GoalForceError("vae_im2col.src must be contiguous [T,H,W,C]")
    T, H, W, C = src.shape
    if cache is not None:
        _req(cache, "vae_im2col.cache")
        if tuple(cache.shape) != (2, H, W, C) or not cache.is_contiguous():
            raise GoalForceError(f"vae_im2col.cache must be contiguous [2,{H},{W},{C}]")
    mode = 1 if upsample2x else (2 if downsample2 else 0)
    if t_out is None:
        t_out = (T - t_off + t_stride - 1) // t_stride
    if t_off + (t_out - 1) * t_stride >= T:
        raise GoalForceError("vae_im2col: temporal window exceeds the source")
    px = H * W * 4 if mode == 1 else ((H // 2) * (W // 2) if mode == 2 else H * W)
    out = torch.empty((t_out * px, kpad), dtype=_BF16, device=src.device)
    _lib.check(_lib.load().gf_vae_im2col(_ptr(src), _ptr(cache), _ptr(out), t_out, H, W, C, kt, ks, mode, t_stride, t_off,
                                         kpad, _stream(src)), "gf_vae_im2col")
    return out


def vae_conv3d(src, cache, w, bias, kt, ks, upsample2x=False, downsample2=False, t_stride=1, t_off=0, t_out=None,
               resid=None, out=None, history_in_front=False):
    """Causal conv as one implicit GEMM (gf_conv3d_bf16): src [T,H,W,C] (+cache [2,H,W,C]) x w [N, kpad] -> [T_out*Ho*Wo, N].
    Same arguments as vae_im2col + gemm(epilogue BIAS / BIAS_RESID); bit-identical to that pair.
    history_in_front: src is frames [2:] of one contiguous [2+T,H,W,C] buffer whose first two frames hold the causal history
    (pass cache=None): the faster pointer-per-row gather."""
    _req(src, "vae_conv3d.src")
    _req(w, "vae_conv3d.w")
    if not src.is_contiguous() or src.dim() != 4:
        raise GoalForceError("vae_conv3d.src must be contiguous [T,H,W,C]")
    T, H, W, C = src.shape
    if kt == 3:
        if history_in_front:
            if cache is not None or src.storage_offset() < 2 * H * W * C:
                raise GoalForceError("vae_conv3d: history_in_front needs cache=None and src = buffer[2:] of a [2+T,H,W,C] buffer")
        elif cache is None:
            raise GoalForceError("vae_conv3d: a temporal kernel needs its 2-frame cache (or history_in_front)")
    if cache is not None:
        _req(cache, "vae_conv3d.cache")
        if tuple(cache.shape) != (2, H, W, C) or not cache.is_contiguous():
            raise GoalForceError(f"vae_conv3d.cache must be contiguous [2,{H},{W},{C}]")
    if w.dim() != 2 or w.stride(1) != 1:
        raise GoalForceError("vae_conv3d.w must be [N, kpad] with contiguous rows")
    mode = 1 if upsample2x else (2 if downsample2 else 0)
    if t_out is None:
        t_out = (T - t_off + t_stride - 1) // t_stride
    px = H * W * 4 if mode == 1 else ((H // 2) * (W // 2) if mode == 2 else H * W)
    n, k = w.shape
    if out is None:
        out = torch.empty((t_out * px, n), dtype=_BF16, device=src.device)
    elif out.dim() != 2 or out.shape != (t_out * px, n) or out.stride(1) != 1:
        raise GoalForceError("vae_conv3d.out must be [T_out*Ho*Wo, N] with contiguous rows")
    if bias is not None:
        _req(bias, "vae_conv3d.bias")
    ldr = 0
    if resid is not None:
        _req(resid, "vae_conv3d.resid")
        if resid.dim() != 2 or resid.shape[0] != t_out * px or resid.stride(1) != 1:
            raise GoalForceError("vae_conv3d.resid must be [T_out*Ho*Wo, >=N] with contiguous rows")
        ldr = resid.stride(0)
    prof = PROFILE_CONV
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().gf_conv3d_bf16(_ptr(src), _ptr(cache), _ptr(w), w.stride(0), _ptr(bias), _ptr(out), out.stride(0),
                                          T, t_out, H, W, C, kt, ks, mode, t_stride, t_off, n, k,
                                          EPI_BIAS if resid is None else EPI_BIAS_RESID, _ptr(resid), ldr, _stream(src)),
               "gf_conv3d_bf16")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, t_out * px, n, C, kt, ks, mode, resid is not None))
    return out


PADDED_CONV_CHANNELS = (192, 384)     # the levels whose 3x3x3 convolutions run on gf_conv3d_padded_bf16


def padded_conv_fits(T, H, W, C, history=True):
    """Whether gf_conv3d_padded_bf16 can take this shape: the zero-bordered buffer is addressed with 32-bit byte offsets (< 4 GiB) and
    its rows with 31-bit indices.  A production tile is far below (0.82 GB at the 192-channel level); an UNTILED 480 x 832 decode
    exceeds it at the 384 -> 192 resample convolution (6.3 GB) — such shapes stay on the implicit GEMM."""
    rows = (T + (2 if history else 0)) * (H + 2) * (W + 2)
    return C in PADDED_CONV_CHANNELS and rows * C * 2 < (1 << 32) and rows < (1 << 31)


def padded_activation(T, H, W, C, device, history=True):
    """A zero-bordered conv input [2 + T, H + 2, W + 2, C] (gf_conv3d_padded_bf16) and two views of it: the two history frames'
    interior [2, H, W, C] and the T frames' interior [T, H, W, C].  history=False (a per-frame 3x3 convolution, kt = 1):
    [T, H + 2, W + 2, C] and (buf, None, interior).  Zeroed once here; producers only ever write the interiors."""
    h = 2 if history else 0
    buf = torch.zeros((T + h, H + 2, W + 2, C), dtype=_BF16, device=device)
    return buf, (buf[:2, 1:H + 1, 1:W + 1] if history else None), buf[h:, 1:H + 1, 1:W + 1]


def vae_upsample2x_padded(x, buf):
    """nearest-exact 2x upsample of the contiguous x [T, H, W, C] into the interior of the zero-bordered `buf` [T, 2H + 2, 2W + 2, C]."""
    _req(x, "vae_upsample2x_padded.x")
    _req(buf, "vae_upsample2x_padded.buf")
    T, H, W, C = x.shape
    if not x.is_contiguous() or not buf.is_contiguous() or tuple(buf.shape) != (T, 2 * H + 2, 2 * W + 2, C):
        raise GoalForceError("vae_upsample2x_padded: x [T,H,W,C] contiguous, buf [T,2H+2,2W+2,C] contiguous")
    _lib.check(_lib.load().gf_vae_upsample2x_padded(_ptr(x), buf[0, 1, 1].data_ptr(), T, H, W, C, _stream(x)), "gf_vae_upsample2x_padded")
    return buf[:, 1:2 * H + 1, 1:2 * W + 1]


def vae_rmsnorm_silu_padded(x, gamma, buf, silu=True):
    """RMS_norm(+SiLU) of the contiguous x [T, H, W, C] into frames [2:] of the zero-bordered buffer `buf` [2 + T, H + 2, W + 2, C]."""
    _req(x, "vae_rmsnorm_silu_padded.x")
    _req(gamma, "vae_rmsnorm_silu_padded.gamma")
    _req(buf, "vae_rmsnorm_silu_padded.buf")
    T, H, W, C = x.shape
    if not x.is_contiguous() or not buf.is_contiguous() or tuple(buf.shape) != (T + 2, H + 2, W + 2, C) or gamma.numel() != C:
        raise GoalForceError("vae_rmsnorm_silu_padded: x [T,H,W,C] contiguous, buf [T+2,H+2,W+2,C] contiguous")
    interior = buf[2, 1, 1]
    _lib.check(_lib.load().gf_vae_rmsnorm_silu_padded(_ptr(x), _ptr(gamma), interior.data_ptr(), T, H, W, C, 1 if silu else 0, _stream(x)),
               "gf_vae_rmsnorm_silu_padded")
    return buf[2:, 1:H + 1, 1:W + 1]


def vae_conv3d_padded(buf, w, bias, resid=None, out=None, kt=3):
    """kt = 3: CausalConv3d 3x3x3 on the zero-bordered activation `buf` [2 + T, H + 2, W + 2, C] (frames 0, 1 = history); kt = 1: a
    3x3 convolution per frame on `buf` [T, H + 2, W + 2, C].  x w [N, >= 9 kt C] -> [T*H*W, N]; resid [T*H*W, >= N] is added after
    the bf16 rounding of conv + bias.  Bit-identical to vae_conv3d."""
    _req(buf, "vae_conv3d_padded.buf")
    _req(w, "vae_conv3d_padded.w")
    if buf.dim() != 4 or not buf.is_contiguous() or kt not in (1, 3):
        raise GoalForceError("vae_conv3d_padded.buf must be contiguous [kt-1+T, H+2, W+2, C], kt = 1 or 3")
    T, H, W, C = buf.shape[0] - (kt - 1), buf.shape[1] - 2, buf.shape[2] - 2, buf.shape[3]
    n = w.shape[0]
    if w.dim() != 2 or w.stride(1) != 1 or w.shape[1] < 9 * kt * C:
        raise GoalForceError("vae_conv3d_padded.w must be [N, >= 9 kt C] with contiguous rows")
    if out is None:
        out = torch.empty((T * H * W, n), dtype=_BF16, device=buf.device)
    elif out.dim() != 2 or out.shape != (T * H * W, n) or out.stride(1) != 1:
        raise GoalForceError("vae_conv3d_padded.out must be [T*H*W, N] with contiguous rows")
    ldr = 0
    if bias is not None:
        _req(bias, "vae_conv3d_padded.bias")
    if resid is not None:
        _req(resid, "vae_conv3d_padded.resid")
        if resid.dim() != 2 or resid.shape[0] != T * H * W or resid.stride(1) != 1:
            raise GoalForceError("vae_conv3d_padded.resid must be [T*H*W, >= N] with contiguous rows")
        ldr = resid.stride(0)
    prof = PROFILE_CONV
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().gf_conv3d_padded_bf16(_ptr(buf), _ptr(w), w.stride(0), _ptr(bias), _ptr(out), out.stride(0), T, H, W, C, n, kt,
                                                 EPI_BIAS if resid is None else EPI_BIAS_RESID, _ptr(resid), ldr, _stream(buf)),
               "gf_conv3d_padded_bf16")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, T * H * W, n, C, kt, 3, 3, resid is not None))      # mode 3 = the padded-layout direct kernel
    return out


def vae_finish_latent(x, mean, inv_std, C):
    """x [rows, >=C] (conv1 output, channels-last) -> [rows, C] normalised mu."""
    _req(x, "vae_finish_latent.x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise GoalForceError("vae_finish_latent.x must be 2-D with contiguous rows")
    out = torch.empty((x.shape[0], C), dtype=_BF16, device=x.device)
    _lib.check(_lib.load().gf_vae_finish_latent(_ptr(x), x.stride(0), _ptr(mean), _ptr(inv_std), _ptr(out), x.shape[0], C,
                                                _stream(x)), "gf_vae_finish_latent")
    return out


def vae_rmsnorm_silu(x, gamma, silu=True, out=None):
    """x [..., C] channels-last -> RMS_norm(+SiLU), same shape (into `out` if given: contiguous, same shape)."""
    _req(x, "vae_rmsnorm_silu.x")
    _req(gamma, "vae_rmsnorm_silu.gamma")
    if not x.is_contiguous() or gamma.numel() != x.shape[-1]:
        raise GoalForceError("vae_rmsnorm_silu: x must be contiguous and gamma match the channel dim")
    if out is None:
        out = torch.empty_like(x)
    else:
        _req(out, "vae_rmsnorm_silu.out")
        if out.shape != x.shape or not out.is_contiguous():
            raise GoalForceError("vae_rmsnorm_silu.out must be contiguous with the shape of x")
    C = x.shape[-1]
    _lib.check(_lib.load().gf_vae_rmsnorm_silu(_ptr(x), _ptr(gamma), _ptr(out), x.numel() // C, C, 1 if silu else 0,
                                               _stream(x)), "gf_vae_rmsnorm_silu")
    return out


def softmax_rows(x, scale, ldo, bias=None, nvalid=None):
    """out[r,:] = softmax(bf16(x[r,:]*scale + bias[r,:])) over the first nvalid columns; zero-padded to ldo columns."""
    _req(x, "softmax_rows.x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise GoalForceError("softmax_rows.x must be 2-D with contiguous rows")
    ldb = 0
    if bias is not None:
        _req(bias, "softmax_rows.bias")
        if bias.shape != x.shape or bias.stride(1) != 1:
            raise GoalForceError("softmax_rows.bias must match x with contiguous rows")
        ldb = bias.stride(0)
    out = torch.empty((x.shape[0], ldo), dtype=_BF16, device=x.device)
    _lib.check(_lib.load().gf_softmax_rows(_ptr(x), x.stride(0), _ptr(bias), ldb, _ptr(out), ldo, x.shape[0], x.shape[1],
                                           x.shape[1] if nvalid is None else int(nvalid), float(scale), _stream(x)),
               "gf_softmax_rows")
    return out


def rowmax_neg(x, out_col):
    """out_col[r] = -max_c x[r, c] (bf16, exact): x [R, C] with contiguous rows, out_col a [R] VIEW (any stride) — a column of the
    augmented Q operand of the VAE attention's second score GEMM (gf_rowmax_neg_bf16)."""
    _req(x, "rowmax_neg.x")
    _req(out_col, "rowmax_neg.out_col")
    if x.dim() != 2 or x.stride(1) != 1 or out_col.dim() != 1 or out_col.shape[0] != x.shape[0]:
        raise GoalForceError("rowmax_neg: x [R, C] with contiguous rows, out_col [R]")
    _lib.check(_lib.load().gf_rowmax_neg_bf16(_ptr(x), x.stride(0), _ptr(out_col), out_col.stride(0) if out_col.shape[0] > 1 else 1,
                                              x.shape[0], x.shape[1], _stream(x)), "gf_rowmax_neg_bf16")
    return out_col


def transpose_pad(x, rpad):
    _req(x, "transpose_pad.x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise GoalForceError("transpose_pad.x must be 2-D with contiguous rows")
    R, C = x.shape
    out = torch.empty((C, rpad), dtype=_BF16, device=x.device)
    _lib.check(_lib.load().gf_transpose_pad(_ptr(x), x.stride(0), _ptr(out), R, C, rpad, _stream(x)), "gf_transpose_pad")
    return out


def transpose_pad_batched(x, rpad):
    """x [B, R, C] (any batch / row strides, contiguous last dim) -> [B, C, rpad] (zero columns past R) in one launch."""
    _req(x, "transpose_pad_batched.x")
    if x.dim() != 3 or x.stride(2) != 1:
        raise GoalForceError("transpose_pad_batched.x must be 3-D with a contiguous last dim")
    B, R, C = x.shape
    out = torch.empty((B, C, rpad), dtype=_BF16, device=x.device)
    _lib.check(_lib.load().gf_transpose_pad_batched(_ptr(x), x.stride(1), x.stride(0), _ptr(out), C * rpad, R, C, rpad, B, _stream(x)),
               "gf_transpose_pad_batched")
    return out


def gemm_batched(a, w, out=None):
    """out[b] = a[b] @ w[b]^T for b < B in ONE launch (gf_gemm_bf16_batched): a [B, M, K], w [B, N, K], out [B, M, N] — views with any
    batch / row strides and a contiguous last dim (heads of a [L, H*d] tensor: t.view(L, H, d).permute(1, 0, 2)).  No bias / epilogue.
    The 8-wave kernel: bit-identical to B calls of gemm() under options(prefer_8wave=1) (for M >= 512 gemm() takes the 4-wave kernel, which sums the
    same products from a rotated K tile on: equal up to fp32 summation order)."""
    _req(a, "gemm_batched.a")
    _req(w, "gemm_batched.w")
    if a.dim() != 3 or w.dim() != 3 or a.stride(2) != 1 or w.stride(2) != 1 or a.shape[0] != w.shape[0] or a.shape[2] != w.shape[2]:
        raise GoalForceError("gemm_batched: a [B, M, K] and w [B, N, K] with contiguous last dims")
    B, M, K = a.shape
    N = w.shape[1]
    if out is None:
        out = torch.empty((B, M, N), dtype=_BF16, device=a.device)
    else:
        _req(out, "gemm_batched.out")
        if tuple(out.shape) != (B, M, N) or out.stride(2) != 1:
            raise GoalForceError(f"gemm_batched.out: expected [{B}, {M}, {N}] with a contiguous last dim")
    _lib.check(_lib.load().gf_gemm_bf16_batched(_ptr(a), a.stride(1), a.stride(0), _ptr(w), w.stride(1), w.stride(0), _ptr(out),
                                                out.stride(1), out.stride(0), M, N, K, B, _stream(a)), "gf_gemm_bf16_batched")
    return out


def vae_tile_blend(values, weight, tile, y0, x0, bounds, border):
    """values [nch,T,H,W], weight [H,W], tile [T,th,tw,tc>=nch] channels-last; bounds=(top,bottom,left,right)."""
    for n, t in (("values", values), ("weight", weight), ("tile", tile)):
        _req(t, f"vae_tile_blend.{n}")
        if not t.is_contiguous():
            raise GoalForceError(f"vae_tile_blend.{n} must be contiguous")
    nch, T, H, W = values.shape
    Tt, th, tw, tc = tile.shape
    if Tt != T or tuple(weight.shape) != (H, W):
        raise GoalForceError("vae_tile_blend: shape mismatch")
    _lib.check(_lib.load().gf_vae_tile_blend(_ptr(values), _ptr(weight), _ptr(tile), nch, T, th, tw, tc, H, W, y0, x0,
                                             int(bounds[0]), int(bounds[1]), int(bounds[2]), int(bounds[3]),
                                             border[0], border[1], _stream(values)), "gf_vae_tile_blend")


def vae_tile_finalize(values, weight, clamp=True):
    _lib.check(_lib.load().gf_vae_tile_finalize(_ptr(values), _ptr(weight), values.shape[0] * values.shape[1],
                                                weight.numel(), 1 if clamp else 0, _stream(values)), "gf_vae_tile_finalize")
    return values


# ---------------------------------------------------------------------------------- fp8 Linear (VRAM:115-151)
_FP8 = torch.float8_e4m3fn


def quant_fp8_rowscale(x):
    """x [M,K] bf16 -> (x8 [M,K] float8_e4m3fn, scale_a [M] fp32) — gf_quant_fp8_rowscale."""
    _req(x, "quant_fp8_rowscale.x")
    xv, rows, dim, xs = _rows2d(x, "quant_fp8_rowscale.x")
    out = torch.empty((rows, dim), dtype=_FP8, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().gf_quant_fp8_rowscale(_ptr(xv), _ptr(out), _ptr(scale), rows, dim, xs, dim, _stream(x)),
               "gf_quant_fp8_rowscale")
    return out, scale


LN_FP8_WIDTHS = (5120, 4096, 1536)     # gf_layernorm_modulate_fp8's wave-per-row widths


def layernorm_modulate_fp8(x, weight=None, bias=None, scale1p=None, shift=None, eps=1e-6):
    """LayerNorm(+affine)(+modulate) straight into the fp8_linear input format: (x8 [M,K] float8_e4m3fn, scale_a [M] fp32) —
    gf_layernorm_modulate_fp8 at the DiT widths (the bf16 row never reaches HBM), elsewhere the two kernels it fuses; the same
    bits either way."""
    _req(x, "layernorm_modulate_fp8.x")
    xv, rows, dim, xs = _rows2d(x, "layernorm_modulate_fp8.x")
    if dim not in LN_FP8_WIDTHS:
        return quant_fp8_rowscale(layernorm_modulate(x, weight=weight, bias=bias, scale1p=scale1p, shift=shift, eps=eps))
    for n, t in (("weight", weight), ("bias", bias), ("scale1p", scale1p), ("shift", shift)):
        if t is not None:
            _req(t, f"layernorm_modulate_fp8.{n}")
            if t.numel() != dim or not t.is_contiguous():
                raise GoalForceError(f"layernorm_modulate_fp8.{n}: expected contiguous [{dim}]")
    out = torch.empty((rows, dim), dtype=_FP8, device=x.device)
    scale = torch.empty((rows,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().gf_layernorm_modulate_fp8(_ptr(xv), _ptr(out), _ptr(scale), _ptr(weight), _ptr(bias), _ptr(scale1p),
                                                     _ptr(shift), rows, dim, xs, dim, float(eps), _stream(x)),
               "gf_layernorm_modulate_fp8")
    return out, scale


def cast_fp8(w):
    """bf16 -> float8_e4m3fn, elementwise (weights; unit scale)."""
    _req(w, "cast_fp8.w")
    if not w.is_contiguous() or w.numel() % 8:
        raise GoalForceError("cast_fp8: contiguous tensor with numel % 8 == 0 expected")
    out = torch.empty(w.shape, dtype=_FP8, device=w.device)
    _lib.check(_lib.load().gf_cast_fp8(_ptr(w), _ptr(out), w.numel(), _stream(w)), "gf_cast_fp8")
    return out


def gemm_fp8(a8, row_scale, w8, bias=None, epilogue=EPI_BIAS, resid=None, gate=None, out=None):
    """out[M,N] = epilogue((a8 @ w8^T) * row_scale[:,None] + bias) — gf_gemm_fp8."""
    _req(a8, "gemm_fp8.a8", _FP8)
    _req(w8, "gemm_fp8.w8", _FP8)
    _req(row_scale, "gemm_fp8.row_scale", torch.float32)
    if a8.dim() != 2 or w8.dim() != 2 or a8.stride(1) != 1 or w8.stride(1) != 1 or a8.shape[1] != w8.shape[1]:
        raise GoalForceError("gemm_fp8: a8 [M,K], w8 [N,K] with contiguous rows expected")
    M, K = a8.shape
    N = w8.shape[0]
    if row_scale.numel() != M or not row_scale.is_contiguous():
        raise GoalForceError("gemm_fp8.row_scale: contiguous [M] expected")
    if out is None:
        out = torch.empty((M, N), dtype=_BF16, device=a8.device)
    ov, Mo, No, ldc = _rows2d(out, "gemm_fp8.out")
    if (Mo, No) != (M, N):
        raise GoalForceError("gemm_fp8.out: shape mismatch")
    ldr = 0
    if resid is not None:
        _req(resid, "gemm_fp8.resid")
        resid, Mr, Nr, ldr = _rows2d(resid, "gemm_fp8.resid")
        if (Mr, Nr) != (M, N):
            raise GoalForceError("gemm_fp8.resid: shape mismatch")
    for n, t in (("gate", gate), ("bias", bias)):
        if t is not None:
            _req(t, f"gemm_fp8.{n}")
            if t.numel() != N or not t.is_contiguous():
                raise GoalForceError(f"gemm_fp8.{n}: expected contiguous [{N}]")
    prof = PROFILE_GEMM
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().gf_gemm_fp8(_ptr(a8), a8.stride(0), _ptr(w8), w8.stride(0), _ptr(row_scale), _ptr(bias),
                                       _ptr(ov), ldc, M, N, K, int(epilogue), _ptr(resid), ldr, _ptr(gate), _stream(a8)),
               "gf_gemm_fp8")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, M, N, K, int(epilogue)))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# backward / training ops (gf_backward.hip, gf_attention_bwd.hip)
def _f32(t, name):
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise GoalForceError(f"{name}: expected a contiguous fp32 CUDA tensor")
    return t


def layernorm_bwd(x, dy, g=None, dg_acc=None, db_acc=None, eps=1e-6):
    """dx of LayerNorm with y = xhat*g (+b) (g = affine weight, or 1+scale, or None); fp32 [dim] accumulators optional."""
    _req(x, "layernorm_bwd.x")
    _req(dy, "layernorm_bwd.dy")
    xv, rows, dim, xs = _rows2d(x, "layernorm_bwd.x")
    dv, _, _, ds = _rows2d(dy, "layernorm_bwd.dy")
    dx = torch.empty((rows, dim), dtype=_BF16, device=x.device)
    if g is not None:
        _req(g, "layernorm_bwd.g")
    _lib.check(_lib.load().gf_layernorm_bwd(_ptr(xv), xs, _ptr(dv), ds, _ptr(g),
                                            _ptr(dx), dim, None if dg_acc is None else _ptr(_f32(dg_acc, "dg_acc")),
                                            None if db_acc is None else _ptr(_f32(db_acc, "db_acc")), rows, dim, float(eps),
                                            _stream(x)), "gf_layernorm_bwd")
    return dx.view(x.shape)


def rmsnorm_rope_bwd(x_pre, dy, weight, cos=None, sin=None, head_dim=128, eps=1e-6, dw_acc=None):
    _req(x_pre, "rmsnorm_rope_bwd.x")
    _req(dy, "rmsnorm_rope_bwd.dy")
    xv, rows, dim, xs = _rows2d(x_pre, "rmsnorm_rope_bwd.x")
    dv, _, _, ds = _rows2d(dy, "rmsnorm_rope_bwd.dy")
    dx = torch.empty((rows, dim), dtype=_BF16, device=x_pre.device)
    _req(weight, "rmsnorm_rope_bwd.weight")
    _lib.check(_lib.load().gf_rmsnorm_rope_bwd(_ptr(xv), xs, _ptr(dv), ds, _ptr(weight),
                                               None if cos is None else _ptr(_f32(cos, "cos")),
                                               None if sin is None else _ptr(_f32(sin, "sin")), _ptr(dx), dim,
                                               None if dw_acc is None else _ptr(_f32(dw_acc, "dw_acc")), rows, dim,
                                               int(head_dim), float(eps), _stream(x_pre)), "gf_rmsnorm_rope_bwd")
    return dx.view(x_pre.shape)


def colsum(a, b=None, gate=None, acc=None):
    """acc[n] += sum_r a[r,n]*(b[r,n] or 1); with `gate` also returns bf16(a*gate) (else None)."""
    _req(a, "colsum.a")
    av, rows, cols, lda = _rows2d(a, "colsum.a")
    ldb, bp = 0, None
    if b is not None:
        _req(b, "colsum.b")
        bv, _, _, ldb = _rows2d(b, "colsum.b")
        bp = _ptr(bv)
    out = None
    if gate is not None:
        _req(gate, "colsum.gate")
        out = torch.empty((rows, cols), dtype=_BF16, device=a.device)
    _lib.check(_lib.load().gf_colsum(_ptr(av), lda, bp, ldb, None if gate is None else _ptr(gate),
                                     None if out is None else _ptr(out), cols, None if acc is None else _ptr(_f32(acc, "acc")),
                                     rows, cols, _stream(a)), "gf_colsum")
    return None if out is None else out.view(a.shape)


def act_bwd(u, df, kind: str):
    _req(u, "act_bwd.u")
    _req(df, "act_bwd.df")
    if not u.is_contiguous() or not df.is_contiguous() or u.numel() != df.numel():
        raise GoalForceError("act_bwd: u/df must be contiguous and equal-sized")
    du = torch.empty_like(u)
    _lib.check(_lib.load().gf_act_bwd(_ptr(u), _ptr(df), _ptr(du), u.numel(), {"gelu_tanh": 0, "silu": 1}[kind], _stream(u)),
               "gf_act_bwd")
    return du


def mse_loss(pred, target, weight=1.0, want_grad=True):
    """(loss fp32 [1] = weight*mean((pred-target)^2), dpred bf16 or None)."""
    _req(pred, "mse_loss.pred")
    _req(target, "mse_loss.target")
    if not pred.is_contiguous() or not target.is_contiguous() or pred.numel() != target.numel():
        raise GoalForceError("mse_loss: pred/target must be contiguous and equal-sized")
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    _lib.check(_lib.load().gf_mse_loss(_ptr(pred), _ptr(target), None if dpred is None else _ptr(dpred), _ptr(loss),
                                       pred.numel(), float(weight), _stream(pred)), "gf_mse_loss")
    return loss, dpred


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    _req(param, "adamw_step.param")
    _req(grad, "adamw_step.grad")
    if not param.is_contiguous() or not grad.is_contiguous() or grad.numel() != param.numel():
        raise GoalForceError("adamw_step: param/grad must be contiguous and equal-sized")
    _lib.check(_lib.load().gf_adamw_step(_ptr(param), _ptr(grad), _ptr(_f32(exp_avg, "exp_avg")),
                                         _ptr(_f32(exp_avg_sq, "exp_avg_sq")), param.numel(), float(lr), float(betas[0]),
                                         float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale),
                                         _stream(param)), "gf_adamw_step")
    # the kernel wrote through the raw pointer: tell torch, so that everything derived from the old values (K-padded
    # patch weights, fp8 copies, the all-zero flag of a ControlNet — dit.param_key) is rebuilt on its next use
    torch.autograd.graph.increment_version(param)


def f32_to_bf16(acc):
    _f32(acc, "f32_to_bf16.acc")
    out = torch.empty(acc.shape, dtype=_BF16, device=acc.device)
    _lib.check(_lib.load().gf_f32_to_bf16(_ptr(acc), _ptr(out), acc.numel(), _stream(acc)), "gf_f32_to_bf16")
    return out


def sumsq(x, acc):
    """acc[0] += sum(x^2) over a contiguous bf16 tensor (fp32 accumulate)."""
    _req(x, "sumsq.x")
    if not x.is_contiguous():
        raise GoalForceError("sumsq.x must be contiguous")
    _lib.check(_lib.load().gf_sumsq(_ptr(x), x.numel(), _ptr(_f32(acc, "sumsq.acc")), _stream(x)), "gf_sumsq")
