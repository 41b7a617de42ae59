"""WanVideoVAE (Wan2.1 VAE, z=16, base dim 96) — decoder path on the HIP kernels.

Drop-in surface: `WanVideoVAE.decode(hidden_states, device, tiled, tile_size,
tile_stride) -> [B,3,T,H,W]` in [-1,1], `.upsampling_factor = 8`, `.model.z_dim = 16`, state_dict keys of the
reference (`model.encoder.*`, `model.conv1.*`, `model.conv2.*`, `model.decoder.*`; diffsynth/models/wan_video_vae.py,
"VAE").  `encode` (used by the pre-loop units GF:791-805, 887-917) runs on the same kernels.

How the reference's algorithm maps here
  * VideoVAE_.decode (VAE:1011-1034) decodes latent frame by latent frame through Decoder3d with a per-conv
    feature cache of the last CACHE_T=2 input frames.  We keep exactly that streaming structure: every
    causal conv owns a [2,H,W,C] cache (all-zero = the reference's `None`, i.e. zero temporal padding) and
    after each chunk stores the last two frames of (cache ++ input) — the same rule as VAE:283-294 incl.
    the `cache_x.shape[2] < 2` branch.
  * the 'Rep' sentinel of upsample3d (VAE:125-153): the first chunk skips time_conv (no temporal
    doubling), so chunk 0 yields 1 frame and every later chunk 4 — reproduced by `first`.
  * activations are channels-last [T,H,W,C]; conv = gf_vae_im2col + gf_gemm_bf16 (weights permuted once to
    [Cout,(dt,dy,dx,cin)]); RMS_norm+SiLU fused; residual adds are GEMM epilogues; nearest-exact 2x upsample is
    folded into the following conv's gather; AttentionBlock = 2 GEMMs + row softmax + GEMM.
  * tiled_decode (VAE:1103-1152): same tile order, ramps and bf16 accumulators, but the blend runs on the GPU
    (the reference moves every tile to the CPU) and the result stays a device tensor.
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import GoalForceError

CACHE_T = 2
VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
            0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]   # VAE:1063-1066
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
           3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]        # VAE:1067-1070


def _conv(x, cache, c, resid=None, **gather):
    """One prepared conv `c` (dict of _prep_conv) on x [T,H,W,C] (+ 2-frame cache) -> [T_out*Ho*Wo, N]: ONE implicit GEMM
    (gf_conv3d_bf16: no patch matrix in HBM).  The patch-matrix form (gf_vae_im2col + gf_gemm_bf16) is the tests' cross-check of
    it (tests/test_vae.py), not a product path."""
    return ops.vae_conv3d(x, cache, c["w"], c["b"], c["kt"], c["ks"], resid=resid, **gather)


def _with_history(shape, like):
    """A [2+T,H,W,C] buffer and its frames [2:]: a conv input written into the second part has room for its two history frames
    directly in front of it (gf_conv3d_bf16's pointer-per-row gather)."""
    T, H, W, C = shape
    buf = torch.empty((T + CACHE_T, H, W, C), dtype=like.dtype, device=like.device)
    return buf, buf[CACHE_T:]


def _roll_cache(cache, x):
    """The feature cache after a chunk: the last CACHE_T frames of [cache, x] (VAE:283-294) ."""
    if x.shape[0] >= CACHE_T:
        return x[-CACHE_T:].clone()       # a copy: a view would keep the whole group of frames alive
    return torch.cat([cache[x.shape[0] - CACHE_T:], x], dim=0)


def _pad_to(n, m):
    return -(-n // m) * m


def decoder_layout(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_upsample=(True, True, False)):
    """Decoder3d structure (VAE:736-786) as a flat list of (kind, name, ...) + parameter shapes."""
    dims = [dim * u for u in [dim_mult[-1]] + list(dim_mult[::-1])]
    shapes: Dict[str, Tuple[int, ...]] = {}
    plan: List[tuple] = []

    def conv3(name, cin, cout, k=(3, 3, 3)):
        shapes[name + ".weight"] = (cout, cin) + tuple(k)
        shapes[name + ".bias"] = (cout,)

    def res(name, cin, cout):
        shapes[f"{name}.residual.0.gamma"] = (cin, 1, 1, 1)
        conv3(f"{name}.residual.2", cin, cout)
        shapes[f"{name}.residual.3.gamma"] = (cout, 1, 1, 1)
        conv3(f"{name}.residual.6", cout, cout)
        if cin != cout:
            conv3(f"{name}.shortcut", cin, cout, (1, 1, 1))
        plan.append(("res", name, cin, cout))

    conv3("decoder.conv1", z_dim, dims[0])
    plan.append(("conv1", "decoder.conv1", z_dim, dims[0]))
    res("decoder.middle.0", dims[0], dims[0])
    shapes["decoder.middle.1.norm.gamma"] = (dims[0], 1, 1)
    shapes["decoder.middle.1.to_qkv.weight"] = (dims[0] * 3, dims[0], 1, 1)
    shapes["decoder.middle.1.to_qkv.bias"] = (dims[0] * 3,)
    shapes["decoder.middle.1.proj.weight"] = (dims[0], dims[0], 1, 1)
    shapes["decoder.middle.1.proj.bias"] = (dims[0],)
    plan.append(("attn", "decoder.middle.1", dims[0]))
    res("decoder.middle.2", dims[0], dims[0])
    idx = 0
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            in_dim = in_dim // 2
        for _ in range(num_res_blocks + 1):
            res(f"decoder.upsamples.{idx}", in_dim, out_dim)
            idx += 1
            in_dim = out_dim
        if i != len(dim_mult) - 1:
            name = f"decoder.upsamples.{idx}"
            shapes[name + ".resample.1.weight"] = (out_dim // 2, out_dim, 3, 3)
            shapes[name + ".resample.1.bias"] = (out_dim // 2,)
            t3 = bool(temperal_upsample[i])
            if t3:
                conv3(name + ".time_conv", out_dim, out_dim * 2, (3, 1, 1))
            plan.append(("up", name, out_dim, t3))
            idx += 1
    shapes["decoder.head.0.gamma"] = (dims[-1], 1, 1, 1)
    conv3("decoder.head.2", dims[-1], 3)
    plan.append(("head", "decoder.head", dims[-1]))
    conv3("conv2", z_dim, z_dim, (1, 1, 1))
    return plan, shapes


def encoder_layout(dim=96, z_dim=16, dim_mult=(1, 2, 4, 4), num_res_blocks=2, temperal_downsample=(False, True, True)):
    """Encoder3d structure (VAE:517-566) + VideoVAE_.conv1 as (plan, parameter shapes)."""
    dims = [dim * u for u in [1] + list(dim_mult)]
    shapes: Dict[str, Tuple[int, ...]] = {}
    plan: List[tuple] = []

    def conv3(name, cin, cout, k=(3, 3, 3)):
        shapes[name + ".weight"] = (cout, cin) + tuple(k)
        shapes[name + ".bias"] = (cout,)

    def res(name, cin, cout):
        shapes[f"{name}.residual.0.gamma"] = (cin, 1, 1, 1)
        conv3(f"{name}.residual.2", cin, cout)
        shapes[f"{name}.residual.3.gamma"] = (cout, 1, 1, 1)
        conv3(f"{name}.residual.6", cout, cout)
        if cin != cout:
            conv3(f"{name}.shortcut", cin, cout, (1, 1, 1))
        plan.append(("res", name, cin, cout))

    conv3("encoder.conv1", 3, dims[0])
    plan.append(("conv1", "encoder.conv1", 3, dims[0]))
    idx = 0
    out_dim = dims[0]
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(num_res_blocks):
            res(f"encoder.downsamples.{idx}", in_dim, out_dim)
            idx += 1
            in_dim = out_dim
        if i != len(dim_mult) - 1:
            name = f"encoder.downsamples.{idx}"
            shapes[name + ".resample.1.weight"] = (out_dim, out_dim, 3, 3)
            shapes[name + ".resample.1.bias"] = (out_dim,)
            t3 = bool(temperal_downsample[i])
            if t3:
                conv3(name + ".time_conv", out_dim, out_dim, (3, 1, 1))
            plan.append(("down", name, out_dim, t3))
            idx += 1
    res("encoder.middle.0", out_dim, out_dim)
    shapes["encoder.middle.1.norm.gamma"] = (out_dim, 1, 1)
    shapes["encoder.middle.1.to_qkv.weight"] = (out_dim * 3, out_dim, 1, 1)
    shapes["encoder.middle.1.to_qkv.bias"] = (out_dim * 3,)
    shapes["encoder.middle.1.proj.weight"] = (out_dim, out_dim, 1, 1)
    shapes["encoder.middle.1.proj.bias"] = (out_dim,)
    plan.append(("attn", "encoder.middle.1", out_dim))
    res("encoder.middle.2", out_dim, out_dim)
    shapes["encoder.head.0.gamma"] = (out_dim, 1, 1, 1)
    conv3("encoder.head.2", out_dim, z_dim * 2)
    plan.append(("head", "encoder.head", out_dim))
    conv3("conv1", z_dim * 2, z_dim * 2, (1, 1, 1))
    return plan, shapes


class _ParamTree(nn.Module):
    """nn.Module whose parameters are registered under dotted names so state_dict() keys equal the
    reference's (e.g. 'decoder.upsamples.4.shortcut.weight')."""

    def put(self, dotted: str, value: torch.Tensor):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _ParamTree())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))

    def get(self, dotted: str) -> torch.Tensor:
        mod = self
        for p in dotted.split("."):
            mod = getattr(mod, p)
        return mod


class VideoVAE_(_ParamTree):
    """Parameter container for VideoVAE_ (VAE:951-977): encoder + conv1 + conv2 + decoder (dim 96, z 16)."""

    def __init__(self, dim=96, z_dim=16):
        super().__init__()
        self.dim, self.z_dim = dim, z_dim
        self.plan, self.shapes = decoder_layout(dim, z_dim)
        self.enc_plan, enc_shapes = encoder_layout(dim, z_dim)
        self.shapes = {**enc_shapes, **self.shapes}
        for name, shp in self.shapes.items():
            if name.endswith("gamma"):
                t = torch.ones(shp)
            elif name.endswith("bias"):
                t = torch.zeros(shp)
            else:
                fan_in = math.prod(shp[1:])
                t = torch.randn(shp) / math.sqrt(fan_in)
            self.put(name, t)


def frame_attention(qkv, C, out):
    """softmax(q k^T / sqrt(C)) v per frame: qkv [g, hw, 3C] bf16 (q | k | v along the last dim), out [g, hw, C] (VAE:326-333).

    Scores: the reference's F.scaled_dot_product_attention (VAE:331) keeps q k^T in fp32; a GEMM with a bf16 output rounds the RAW
    scores (2^-9 of their magnitude: 1e-2 on the block's output at logit std 3, where SDPA sits at 1.8e-3).  The softmax only needs
    s - (row constant), so: a first, coarse GEMM -> minus its row maximum into an extra K column of Q (gf_rowmax_neg_bf16; K's extra
    column is 1) -> the second GEMM accumulates q.k - max in fp32 and rounds a number that is small where the softmax weight is large.
    ops.options(vae_attn_offset=False) is the one-GEMM form (tests/test_vae.py measures both against fp64)."""
    g, hw, _ = qkv.shape
    kp = _pad_to(hw, 64)
    q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    scores = ops.gemm_batched(q, k)                                                # q k^T  [g, hw, hw]
    if ops._OPT["vae_attn_offset"]:
        qa = torch.zeros((g, hw, C + 64), dtype=qkv.dtype, device=qkv.device)      # [q, -rowmax, 0 ...]
        ka = torch.zeros((g, hw, C + 64), dtype=qkv.dtype, device=qkv.device)      # [k,    1,    0 ...]
        qa[:, :, :C].copy_(q)
        ka[:, :, :C].copy_(k)
        ka[:, :, C] = 1
        ops.rowmax_neg(scores.view(g * hw, hw), qa.view(g * hw, C + 64)[:, C])
        ops.gemm_batched(qa, ka, out=scores)                                       # q k^T - rowmax, rounded once
    p = ops.softmax_rows(scores.view(g * hw, hw), 1.0 / math.sqrt(C), kp)          # [g hw, kp]
    vt = ops.transpose_pad_batched(v, kp)                                          # [g, C, kp]
    return ops.gemm_batched(p.view(g, hw, kp), vt, out=out)                        # [g, hw, C]


class WanVideoVAE(nn.Module):
    def __init__(self, z_dim=16):
        super().__init__()
        self.mean = torch.tensor(VAE_MEAN)
        self.std = torch.tensor(VAE_STD)
        self.scale = [self.mean, 1.0 / self.std]
        self.model = VideoVAE_(z_dim=z_dim)
        self.upsampling_factor = 8
        self.z_dim = z_dim
        self._prepared = None     # GEMM-ready weights (built lazily, invalidated on load)
        self._cache: Dict[str, torch.Tensor] = {}
        self._pad_pool: Dict[tuple, list] = {}    # zero-bordered conv inputs by shape (_padded_buffer); lives inside a _pool_scope
        self._pool_depth, self._pool_sig, self._pool_stream = 0, None, None
        # latent frames per decoder call / 4-frame chunks per encoder call after the first frame (1 = the reference's streaming
        # granularity; any value gives the same bits).  20 = the whole 81-frame clip of a tile: ~10 GB of activations.
        # Plain attributes (no environment variable reads them): tests and tools assign them.
        self.frames_per_chunk = 20
        # the clip's first frame in one chunk with the first group of frames (False: a chunk of its own, as the reference streams it)
        self.merge_first = True

    # ---------------------------------------------------------------- state dict
    def load_state_dict(self, state_dict, strict=True, **kw):
        self._prepared = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _apply(self, fn, *a, **k):
        self._prepared = None
        return super()._apply(fn, *a, **k)

    # ---------------------------------------------------------------- weight preparation
    def _prep_conv(self, name, cout_pad_to=8):
        w = self.model.get(name + ".weight")
        b = self.model.get(name + ".bias")
        cout, cin = w.shape[0], w.shape[1]
        if cin % 8:  # RGB input is stored channels-last padded to 8 channels
            cpad = _pad_to(cin, 8)
            wz = torch.zeros((cout, cpad) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
            wz[:, :cin] = w
            w, cin = wz, cpad
        if w.dim() == 5:
            kt, ks = w.shape[2], w.shape[3]
            wm = w.permute(0, 2, 3, 4, 1).reshape(cout, -1)
        else:
            kt, ks = 1, w.shape[2]
            wm = w.permute(0, 2, 3, 1).reshape(cout, -1)
        k = wm.shape[1]
        kpad, npad = _pad_to(k, 64), _pad_to(cout, cout_pad_to)
        wp = torch.zeros((npad, kpad), dtype=w.dtype, device=w.device)
        wp[:cout, :k] = wm
        bp = torch.zeros((npad,), dtype=b.dtype, device=b.device)
        bp[:cout] = b
        return dict(w=wp.contiguous(), b=bp.contiguous(), kt=kt, ks=ks, cin=cin, cout=cout, kpad=kpad)

    def _prepare(self):
        if self._prepared is not None:
            return self._prepared
        dev = self.model.get("conv2.weight").device
        if dev.type != "cuda" or self.model.get("conv2.weight").dtype != torch.bfloat16:
            raise GoalForceError("WanVideoVAE must be moved to the GPU in bf16 before decode (no CPU fallback exists)")
        P = {}
        for name in self.model.shapes:
            if name.endswith(".weight"):
                base = name[:-len(".weight")]
                P[base] = self._prep_conv(base, cout_pad_to=64 if base == "encoder.head.2" else 8)
        # conv2 consumes the 64-channel padded latent: K padded from 16 to 64 already by _prep_conv
        for name in self.model.shapes:
            if name.endswith("gamma"):
                P[name] = self.model.get(name).reshape(-1).contiguous()
        P["mean"] = self.mean.to(dev, torch.bfloat16)
        P["inv_std"] = (1.0 / self.std).to(dev, torch.bfloat16)
        P["zero3"] = torch.zeros(3, dtype=torch.bfloat16, device=dev)
        P["one3"] = torch.ones(3, dtype=torch.bfloat16, device=dev)
        self._prepared = P
        return P

    # ---------------------------------------------------------------- building blocks (channels-last [T,H,W,C])
    def _causal_conv(self, P, name, x, resid=None, front=None):
        """CausalConv3d with its feature cache (VAE:33-52 + the cache handling of VAE:283-294).  `front`: the [2+T,H,W,C] buffer
        whose frames [2:] are x (see _with_history) — the cached history is copied into its first two frames."""
        c = P[name]
        T, H, W, C = x.shape
        r2 = None if resid is None else resid.reshape(T * H * W, -1)
        cache = None
        if c["kt"] == 3:
            cache = self._cache.get(name)
            if front is not None:
                if x.data_ptr() != front[CACHE_T:].data_ptr() or front.shape[1:] != x.shape[1:]:
                    raise GoalForceError("_causal_conv: x must be frames [2:] of `front`")
                if cache is None:
                    front[:CACHE_T].zero_()
                    self._cache[name] = _roll_cache(front[:CACHE_T], x)
                else:
                    front[:CACHE_T].copy_(cache)
                    self._cache[name] = _roll_cache(cache, x)
                return ops.vae_conv3d(x, None, c["w"], c["b"], 3, c["ks"], resid=r2, history_in_front=True).view(T, H, W, -1)
            if cache is None:
                cache = torch.zeros((CACHE_T, H, W, C), dtype=x.dtype, device=x.device)
            self._cache[name] = _roll_cache(cache, x)
        elif c["ks"] == 1 and c["kpad"] == C:
            return ops.gemm(x.reshape(T * H * W, C), c["w"], c["b"],
                            epilogue=ops.EPI_BIAS if resid is None else ops.EPI_BIAS_RESID, resid=r2).view(T, H, W, -1)
        return _conv(x, cache, c, resid=r2).view(T, H, W, -1)

    def _padded_conv(self, P, name, x, gamma, resid=None):
        """RMS_norm + SiLU + CausalConv3d 3x3x3 of a 192- / 384-channel level (VAE:267-301, 33-52) through the zero-bordered layout of
        gf_conv3d_padded_bf16: the norm writes the interior of the padded buffer, the two cached history frames are copied into the
        interior of its first two frames (zeros without a history), the convolution reads taps as constant row shifts.  Same values,
        same sums as _causal_conv: bit-identical."""
        c = P[name]
        T, H, W, C = x.shape
        buf, hist, _ = self._padded_buffer(T, H, W, C, x.device)
        cur = ops.vae_rmsnorm_silu_padded(x, gamma, buf, silu=True)
        cache = self._cache.get(name)
        if cache is None:
            hist.zero_()
            cache = torch.zeros((CACHE_T, H, W, C), dtype=x.dtype, device=x.device)
        else:
            hist.copy_(cache)
        # the feature cache after the chunk: the last two frames of [cache, x] (VAE:283-294), as a contiguous copy
        if T >= CACHE_T:
            nxt = torch.empty((CACHE_T, H, W, C), dtype=x.dtype, device=x.device)
            nxt.copy_(cur[-CACHE_T:])
        else:
            nxt = torch.cat([cache[T - CACHE_T:], cur.contiguous()], dim=0)
        self._cache[name] = nxt
        r2 = None if resid is None else resid.reshape(T * H * W, -1)
        return ops.vae_conv3d_padded(buf, c["w"], c["b"], resid=r2).view(T, H, W, -1)

    def _padded_buffer(self, T, H, W, C, device, history=True):
        """One zero-bordered buffer per shape: zeroed when created, afterwards only its interior is written.  One is enough — every
        producer and consumer of it runs on the same stream (checked), and a residual block's second conv input is built after the
        first convolution has read the buffer.  The pool lives for one _pool_scope (a decode / encode call, or one tile when the
        tile-level entry points are called directly) and holds the buffers of ONE tile shape at a time (~3 GB at the production tile)."""
        if self._pool_depth <= 0:
            raise GoalForceError("_padded_buffer outside a _pool_scope: the padded-layout buffers would never be released")
        stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else None
        if self._pool_stream is None:
            self._pool_stream = stream
        elif stream != self._pool_stream:
            raise GoalForceError("the VAE's padded-layout buffers are shared by every convolution of a tile: run a decode / encode on "
                                 "ONE stream (the pool was created on another stream than the current one)")
        key = (T, H, W, C, str(device), bool(history))
        if key not in self._pad_pool:
            self._pad_pool[key] = ops.padded_activation(T, H, W, C, device) if history else \
                ops.padded_activation(T, H, W, C, device, history=False)
        return self._pad_pool[key]

    @contextlib.contextmanager
    def _pool_scope(self, tile_sig=None):
        """The lifetime of the padded-layout pool: the OUTERMOST scope drops it on the way out — also when an exception passes
        through (callers of the tile-level entry points never released it).  `tile_sig` (a tile-level entry): a tile
        of another shape than the previous one drops the previous shape's buffers first, so that a decode with ragged edge tiles
        holds one shape's set, not one set per distinct shape."""
        if tile_sig is not None and tile_sig != self._pool_sig:
            self._pad_pool, self._pool_sig = {}, tile_sig
        self._pool_depth += 1
        try:
            yield
        finally:
            self._pool_depth -= 1
            if self._pool_depth == 0:
                self._pad_pool, self._pool_sig, self._pool_stream = {}, None, None

    def _res_block(self, P, name, x, cin, cout):
        """ResidualBlock (VAE:267-301)."""
        h = x if cin == cout else self._causal_conv(P, name + ".shortcut", x)
        # per convolution: the padded-layout kernel where the INPUT has 192 / 384 channels (conv 1: cin, conv 2: cout)
        pad_ok = ops._OPT["conv_padded"] and x.is_contiguous()
        if pad_ok and ops.padded_conv_fits(*x.shape[:3], cin):
            y = self._padded_conv(P, name + ".residual.2", x, P[name + ".residual.0.gamma"])
        else:
            front, y = _with_history(x.shape, x)
            ops.vae_rmsnorm_silu(x, P[name + ".residual.0.gamma"], silu=True, out=y)
            y = self._causal_conv(P, name + ".residual.2", y, front=front)
        if pad_ok and ops.padded_conv_fits(*y.shape[:3], cout):
            return self._padded_conv(P, name + ".residual.6", y, P[name + ".residual.3.gamma"], resid=h)
        front, y2 = _with_history(y.shape, y)
        ops.vae_rmsnorm_silu(y, P[name + ".residual.3.gamma"], silu=True, out=y2)
        return self._causal_conv(P, name + ".residual.6", y2, resid=h, front=front)

    def _attention(self, P, name, x, C):
        """AttentionBlock (VAE:304-342): single head of width C over the h*w positions of each frame."""
        T, H, W, _ = x.shape
        hw = H * W
        if hw % 8:
            raise GoalForceError("VAE attention needs h*w to be a multiple of 8")
        xn = ops.vae_rmsnorm_silu(x, P[name + ".norm.gamma"], silu=False)
        qkv_c, proj_c = P[name + ".to_qkv"], P[name + ".proj"]
        # the two 1x1 convolutions (to_qkv, proj + residual) for all frames of the chunk in one GEMM each; the attention itself per frame
        qkv_all = ops.gemm(xn.reshape(T * hw, C), qkv_c["w"], qkv_c["b"]).view(T, hw, 3 * C)
        o_all = torch.empty((T, hw, C), dtype=x.dtype, device=x.device)
        # the frames' attention in groups (one launch per product and group; a group's scores stay below ~1 GB)
        gsz = max(1, min(T, (1 << 29) // (hw * hw)))
        for t0 in range(0, T, gsz):
            frame_attention(qkv_all[t0:t0 + gsz], C, o_all[t0:t0 + gsz])
        out = ops.gemm(o_all.view(T * hw, C), proj_c["w"], proj_c["b"], epilogue=ops.EPI_BIAS_RESID, resid=x.reshape(T * hw, C))
        return out.view(T, H, W, C)

    def _upsample(self, P, name, x, C, temporal, first):
        """Resample upsample2d / upsample3d (VAE:82-174).  `first`: x starts with the clip's FIRST frame, which skips the temporal
        doubling (VAE:139-147); frames behind it in the same chunk go through time_conv with zero history ('Rep'), exactly as they
        would in a chunk of their own."""
        T, H, W, _ = x.shape
        if temporal:
            x0, xr = (x[:1], x[1:]) if first else (None, x)
            Tr = xr.shape[0]
            if Tr > 0:
                tc = P[name + ".time_conv"]
                cache = self._cache.get(name + ".time_conv")
                if cache is None:  # 'Rep': the first executed time_conv sees zero history (VAE:139-147)
                    cache = torch.zeros((CACHE_T, H, W, C), dtype=x.dtype, device=x.device)
                self._cache[name + ".time_conv"] = _roll_cache(cache, xr)
                y2 = _conv(xr, cache, tc)                                 # [Tr*hw, 2C]
                # channels [jC,(j+1)C) of frame t become frame 2t+j (VAE:155-158); the first frame, if here, in front
                n0 = 0 if x0 is None else 1
                out = torch.empty((n0 + 2 * Tr, H, W, C), dtype=x.dtype, device=x.device)
                if n0:
                    out[0].copy_(x0[0])
                out[n0:].view(Tr, 2, H * W, C).copy_(y2.view(Tr, H * W, 2, C).transpose(1, 2))
                x = out
                T = n0 + 2 * Tr
        rc = P[name + ".resample.1"]
        if ops._OPT["conv_padded"] and rc["w"].shape[0] >= 192 and x.is_contiguous() \
                and ops.padded_conv_fits(T, 2 * H, 2 * W, C, history=False):
            # 384 -> 192: the upsampled frames go into the zero-bordered layout once (4 x the source, one pass) and the 3x3
            # convolution reads its taps as row shifts (gf_conv3d_padded_bf16, kt = 1) — same values, same sums as the folded gather
            up = self._padded_buffer(T, 2 * H, 2 * W, C, x.device, history=False)[0]
            ops.vae_upsample2x_padded(x, up)
            return ops.vae_conv3d_padded(up, rc["w"], rc["b"], kt=1).view(T, 2 * H, 2 * W, -1)
        return _conv(x, None, rc, upsample2x=True).view(T, 2 * H, 2 * W, -1)

    def _decode_chunk(self, P, x, first):
        """Decoder3d.forward on one latent frame (VAE:788-838).  x [1,h,w,16] -> [1 or 4, 8h, 8w, 8]."""
        for step in self.model.plan:
            kind, name = step[0], step[1]
            if kind == "conv1":
                x = self._causal_conv(P, name, x)
            elif kind == "res":
                x = self._res_block(P, name, x, step[2], step[3])
            elif kind == "attn":
                x = self._attention(P, name, x, step[2])
            elif kind == "up":
                x = self._upsample(P, name, x, step[2], step[3], first)
            elif kind == "head":
                front, y = _with_history(x.shape, x)
                ops.vae_rmsnorm_silu(x, P[name + ".0.gamma"], silu=True, out=y)
                x = self._causal_conv(P, name + ".2", y, front=front)
        return x

    def decode_tile_channels_last(self, z_slice: torch.Tensor) -> torch.Tensor:
        """VideoVAE_.decode (VAE:1011-1034) for one [16,T,h,w] latent slice -> [4T-3, 8h, 8w, 8] (RGB in 0..2)."""
        with self._pool_scope(("dec",) + tuple(z_slice.shape)):
            try:
                return self._decode_tile(z_slice)
            finally:
                self._cache = {}          # the feature caches never outlive a tile, whatever happened inside it

    def _decode_tile(self, z_slice):
        P = self._prepare()
        self._cache = {}
        zc = ops.vae_prep_latent(z_slice, P["mean"], P["inv_std"], cpad=64)         # z / (1/std) + mean
        T, h, w, _ = zc.shape
        c2 = P["conv2"]
        x = ops.gemm(zc.reshape(T * h * w, 64), c2["w"], c2["b"]).view(T, h, w, -1)  # conv2, 1x1x1
        # The reference streams one latent frame per call through the feature caches (VAE:1021-1033).  A causal conv over a
        # GROUP of frames with the same 2-frame cache in front computes exactly the same sums, so the frames go through in groups:
        # launches that fill the chip, bit-identical output.  The first frame skips the temporal upsampling; it travels in front of
        # the first group (merge_first) — alone it costs ~60 ms of launches that each fill a twentieth of the chip.
        g = max(1, int(self.frames_per_chunk))
        n0 = 1 + (g if self.merge_first else 0)
        frames = [self._decode_chunk(P, x[:n0].contiguous(), first=True)]
        for i in range(n0, T, g):
            frames.append(self._decode_chunk(P, x[i:i + g].contiguous(), first=False))
        self._cache = {}
        return torch.cat(frames, dim=0)

    # ---------------------------------------------------------------- encoder (VAE:517-617, 988-1010)
    def _downsample(self, P, name, x, C, temporal):
        """Resample downsample2d / downsample3d (VAE:101-112, 159-174)."""
        T, H, W, _ = x.shape
        x = _conv(x, None, P[name + ".resample.1"], downsample2=True).view(T, H // 2, W // 2, -1)
        if temporal:
            key = name + ".time_conv"
            prev = self._cache.get(key)
            x0 = None
            if prev is None:   # the clip's first frame: remembered, no temporal conv (VAE:162-164); frames behind it in this chunk follow below
                prev = torch.cat([torch.zeros_like(x[:1]), x[:1]], dim=0).contiguous()
                x0, x = x[:1], x[1:]
                T -= 1
            if T > 0:          # conv over [prev_last, x_0..x_{T-1}], kernel 3, stride 2, no padding (VAE:166-170)
                self._cache[key] = _roll_cache(prev, x)
                x = _conv(x, prev, P[key], t_stride=2, t_off=1, t_out=T // 2).view(T // 2, H // 2, W // 2, -1)
                if x0 is not None:
                    x = torch.cat([x0, x], dim=0)
            else:
                self._cache[key] = prev
                x = x0
        return x

    def _encode_chunk(self, P, x):
        """Encoder3d.forward on one chunk of frames (VAE:568-617).  x [T,H,W,8] -> [T', H/8, W/8, 64 (32 used)]."""
        for step in self.model.enc_plan:
            kind, name = step[0], step[1]
            if kind == "conv1":
                x = self._causal_conv(P, name, x)
            elif kind == "res":
                x = self._res_block(P, name, x, step[2], step[3])
            elif kind == "attn":
                x = self._attention(P, name, x, step[2])
            elif kind == "down":
                x = self._downsample(P, name, x, step[2], step[3])
            elif kind == "head":
                front, y = _with_history(x.shape, x)
                ops.vae_rmsnorm_silu(x, P[name + ".0.gamma"], silu=True, out=y)
                x = self._causal_conv(P, name + ".2", y, front=front)
        return x

    def encode_tile_channels_last(self, video_slice: torch.Tensor) -> torch.Tensor:
        """VideoVAE_.encode (VAE:988-1010) for one [3,T,H,W] slice (any strides) -> normalised mu [T', H/8, W/8, 16]."""
        with self._pool_scope(("enc",) + tuple(video_slice.shape)):
            try:
                return self._encode_tile(video_slice)
            finally:
                self._cache = {}

    def _encode_tile(self, video_slice):
        P = self._prepare()
        self._cache = {}
        T = video_slice.shape[1]
        xin = ops.vae_prep_latent(video_slice, P["zero3"], P["one3"], cpad=8)    # channels-last, RGB padded to 8
        # frame 0 alone, then the 4-frame chunks of VAE:994-1001 in groups (same sums as one chunk at a time, see decode)
        # (the first frame travels in front of the first group, see decode_tile_channels_last)
        n4 = 4 * ((T - 1) // 4)
        g = 4 * max(1, int(self.frames_per_chunk))
        n0 = 1 + (min(g, n4) if self.merge_first else 0)
        outs = [self._encode_chunk(P, xin[:n0].contiguous())]
        for i in range(n0, 1 + n4, g):
            outs.append(self._encode_chunk(P, xin[i:min(i + g, 1 + n4)].contiguous()))
        self._cache = {}
        h = torch.cat(outs, dim=0)                                               # [T', h, w, 64]
        Tl, hh, ww, _ = h.shape
        c1 = P["conv1"]
        y = ops.gemm(h.reshape(Tl * hh * ww, -1), c1["w"], c1["b"])              # 1x1x1 conv1, [.., 32] = (mu, log_var)
        mu = ops.vae_finish_latent(y, P["mean"], P["inv_std"], self.z_dim)       # (mu - mean) * (1/std)
        return mu.view(Tl, hh, ww, self.z_dim)

    def tiled_encode(self, video, device, tile_size, tile_stride):
        """WanVideoVAE.tiled_encode (VAE:1155-1203); sizes in pixels; video [1,3,T,H,W]."""
        _, _, T, H, W = video.shape
        size_h, size_w = tile_size
        stride_h, stride_w = tile_stride
        tasks = []
        for h in range(0, H, stride_h):
            if h - stride_h >= 0 and h - stride_h + size_h >= H:
                continue
            for w in range(0, W, stride_w):
                if w - stride_w >= 0 and w - stride_w + size_w >= W:
                    continue
                tasks.append((h, h + size_h, w, w + size_w))
        u = self.upsampling_factor
        out_T = (T + 3) // 4
        values = torch.zeros((self.z_dim, out_T, H // u, W // u), dtype=torch.bfloat16, device=video.device)
        weight = torch.zeros((H // u, W // u), dtype=torch.bfloat16, device=video.device)
        for (h, h_, w, w_) in tasks:
            tile = self.encode_tile_channels_last(video[0, :, :, h:h_, w:w_])
            ops.vae_tile_blend(values, weight, tile.contiguous(), h // u, w // u,
                               bounds=(h == 0, h_ >= H, w == 0, w_ >= W),
                               border=((size_h - stride_h) // u, (size_w - stride_w) // u))
        ops.vae_tile_finalize(values, weight, clamp=False)
        return values.unsqueeze(0)

    def single_encode(self, video, device):
        tile = self.encode_tile_channels_last(video[0])
        Tl, hh, ww, _ = tile.shape
        values = torch.zeros((self.z_dim, Tl, hh, ww), dtype=torch.bfloat16, device=video.device)
        weight = torch.zeros((hh, ww), dtype=torch.bfloat16, device=video.device)
        ops.vae_tile_blend(values, weight, tile.contiguous(), 0, 0, bounds=(True, True, True, True), border=(1, 1))
        ops.vae_tile_finalize(values, weight, clamp=False)
        return values.unsqueeze(0)

    @torch.no_grad()
    def encode(self, videos, device=None, tiled=False, tile_size=(34, 34), tile_stride=(18, 16)):
        """videos: iterable of [3,T,H,W] bf16 in [-1,1] (any strides) -> [B,16,(T+3)//4,H/8,W/8] normalised latents
        (VAE:1218-1232).  Tile sizes are given in latent units like the reference and scaled by 8 here."""
        u = self.upsampling_factor
        outs = []
        with self._pool_scope():
            for v in videos:
                v = v.unsqueeze(0)
                if not v.is_cuda:
                    v = v.to(device or "cuda")
                if tiled:
                    z = self.tiled_encode(v, device, (tile_size[0] * u, tile_size[1] * u), (tile_stride[0] * u, tile_stride[1] * u))
                else:
                    z = self.single_encode(v, device)
                outs.append(z.squeeze(0))
        return torch.stack(outs)

    # ---------------------------------------------------------------- public API (VAE:1103-1152, 1211-1247)
    def tiled_decode(self, hidden_states, device, tile_size, tile_stride, tile_group=None):
        """VAE:1103-1152.  `tile_group` (a torch.distributed group whose ranks all hold the same latents, e.g. the two ranks
        of a CFG pair): the tiles are decoded round-robin over the group's ranks, exchanged, and blended by EVERY rank in
        the reference's task order — the blend arithmetic (bf16 accumulators) is order-dependent, so the result is
        bit-identical to the one-GPU decode on all ranks."""
        _, _, T, H, W = hidden_states.shape
        size_h, size_w = tile_size
        stride_h, stride_w = tile_stride
        tasks = []
        for h in range(0, H, stride_h):
            if h - stride_h >= 0 and h - stride_h + size_h >= H:
                continue
            for w in range(0, W, stride_w):
                if w - stride_w >= 0 and w - stride_w + size_w >= W:
                    continue
                tasks.append((h, h + size_h, w, w + size_w))
        u = self.upsampling_factor
        out_T = T * 4 - 3
        values = torch.zeros((3, out_T, H * u, W * u), dtype=torch.bfloat16, device=hidden_states.device)
        weight = torch.zeros((H * u, W * u), dtype=torch.bfloat16, device=hidden_states.device)
        gsize = 1
        if tile_group is not None:
            import torch.distributed as dist
            gsize, grank = dist.get_world_size(tile_group), dist.get_rank(tile_group)
        mine = {}
        if gsize > 1:      # decode my share first (no waiting on the partner), then exchange in task order
            for i, (h, h_, w, w_) in enumerate(tasks):
                if i % gsize == grank:
                    mine[i] = self.decode_tile_channels_last(hidden_states[0, :, :, h:h_, w:w_]).contiguous()
            tc = 8                                             # channels of a decoded tile (RGB in 0..2), decode_tile_channels_last
            assert all(t.shape[-1] == tc for t in mine.values())
        for i, (h, h_, w, w_) in enumerate(tasks):
            if gsize == 1:
                tile = self.decode_tile_channels_last(hidden_states[0, :, :, h:h_, w:w_]).contiguous()
            else:
                owner = i % gsize
                tile = mine.pop(i) if owner == grank else torch.empty(
                    (out_T, (min(h_, H) - h) * u, (min(w_, W) - w) * u, tc), dtype=torch.bfloat16, device=hidden_states.device)
                dist.broadcast(tile, src=dist.get_global_rank(tile_group, owner), group=tile_group)
            ops.vae_tile_blend(values, weight, tile, h * u, w * u,
                               bounds=(h == 0, h_ >= H, w == 0, w_ >= W),
                               border=((size_h - stride_h) * u, (size_w - stride_w) * u))
        ops.vae_tile_finalize(values, weight)
        return values.unsqueeze(0)

    def single_decode(self, hidden_state, device):
        _, _, T, H, W = hidden_state.shape
        u = self.upsampling_factor
        tile = self.decode_tile_channels_last(hidden_state[0])
        values = torch.zeros((3, T * 4 - 3, H * u, W * u), dtype=torch.bfloat16, device=hidden_state.device)
        weight = torch.zeros((H * u, W * u), dtype=torch.bfloat16, device=hidden_state.device)
        ops.vae_tile_blend(values, weight, tile.contiguous(), 0, 0, bounds=(True, True, True, True), border=(1, 1))
        ops.vae_tile_finalize(values, weight)   # mask == 1 everywhere: clamp(tile, -1, 1)
        return values.unsqueeze(0)

    @torch.no_grad()
    def decode(self, hidden_states, device=None, tiled=False, tile_size=(34, 34), tile_stride=(18, 16), tile_group=None):
        """[B,16,T,h,w] bf16 latents -> [B,3,4T-3,8h,8w] bf16 in [-1,1] (device tensor).  tile_group: see tiled_decode."""
        videos = []
        with self._pool_scope():
            for hs in hidden_states:
                hs = hs.unsqueeze(0)
                if not hs.is_cuda:
                    hs = hs.to(device or "cuda")
                v = (self.tiled_decode(hs, device, tile_size, tile_stride, tile_group=tile_group) if tiled
                     else self.single_decode(hs, device))
                videos.append(v.squeeze(0))
        return torch.stack(videos)

