"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A functional (state-dict driven) torch-CPU restatement of the reference's Goal-Force denoising
path.  It exists to CHECK the HIP product path; nothing under goal_force_amd/ may import it.
Only tests/, __graft_entry__.smoke() and bench.py's two BASELINE legs use it: `cpu_baseline` (this graph timed on the host cores) and
`gpu_eager_yardstick` (the same graph on bf16 device tensors through torch-ROCm's own kernels — what the reference's code path would run on
the same GPU) — both reported beside `value`, never inside it.

Pinning: every function here is checked against golden vectors generated in the build container
by importing the reference's own modules (tests/golden/make_goldens.py -> tests/golden/*.npz;
tests/test_oracle_goldens.py).  Trained-weight parity is unpinned (no checkpoints, no network).

Reference files (relative to the reference tree):
  DIT = diffsynth/models/wan_video_dit.py      GF = src/goal_force/wan_video_new.py
  FM  = diffsynth/schedulers/flow_match.py     VRAM = diffsynth/vram_management/layers.py

Two arithmetic modes:
  * dtype=torch.bfloat16 : the reference's own eager arithmetic (bf16 storage, per-op rounding,
    fp32 norms, fp64 RoPE) — what the reference computes on CPU, bit for bit.
  * dtype=torch.float32  : the same graph in fp32 on the (bf16-valued) weights — the "fp32-math"
    oracle the tolerances in SURVEY.md §8(d) are stated against.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ embeddings / RoPE tables
def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    """DIT:68-72 — cat(cos, sin)(t * 10000^(-i/(dim/2))) in fp64, cast back to position.dtype."""
    half = dim // 2
    inv = torch.pow(10000.0, -torch.arange(half, dtype=torch.float64) / half).to(position.device)   # table built on the host as DIT:69
    ang = position.to(torch.float64)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1).to(position.dtype)


def rope_freqs_1d(dim: int, end: int = 1024, theta: float = 10000.0) -> torch.Tensor:
    """DIT:83-89 — complex128 e^{i * pos * theta^(-2k/dim)}, shape [end, dim/2]."""
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].double() / dim))
    ang = torch.outer(torch.arange(end), inv)
    return torch.polar(torch.ones_like(ang), ang)


def rope_freqs_3d(head_dim: int, f: int, h: int, w: int) -> torch.Tensor:
    """DIT:75-80 + GF:1474-1478 — per-token complex table [f*h*w, head_dim/2]: the head's complex
    pairs are split frame / height / width as (d - 2*(d//3)) / (d//3) / (d//3) real dims."""
    df = head_dim - 2 * (head_dim // 3)
    dh = dw = head_dim // 3
    ff, fh, fw = rope_freqs_1d(df), rope_freqs_1d(dh), rope_freqs_1d(dw)
    tab = torch.cat([
        ff[:f].view(f, 1, 1, -1).expand(f, h, w, -1),
        fh[:h].view(1, h, 1, -1).expand(f, h, w, -1),
        fw[:w].view(1, 1, w, -1).expand(f, h, w, -1),
    ], dim=-1)
    return tab.reshape(f * h * w, -1)


def rope_apply(x: torch.Tensor, freqs: torch.Tensor, num_heads: int) -> torch.Tensor:
    """DIT:92-97 — complex multiply of adjacent pairs in fp64, cast back.  x [B,S,H*d], freqs [S,d/2]."""
    b, s, _ = x.shape
    xc = torch.view_as_complex(x.to(torch.float64).reshape(b, s, num_heads, -1, 2))
    y = torch.view_as_real(xc * freqs[None, :, None, :]).flatten(2)
    return y.to(x.dtype)


# ------------------------------------------------------------------ norms
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """DIT:100-111 — over the FULL last dim, fp32 math, cast to x.dtype, then * weight."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return y.to(x.dtype) * weight


def layer_norm(x: torch.Tensor, weight=None, bias=None, eps: float = 1e-6) -> torch.Tensor:
    """VRAM:78-92 (WanAutoCastLayerNorm) — fp32 LayerNorm, affine params upcast, one cast back.
    (The plain nn.LayerNorm of DIT:206-208 on bf16 input differs by < 1 bf16 ulp.)"""
    w = None if weight is None else weight.float()
    b = None if bias is None else bias.float()
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps).type_as(x)


def modulate(x, shift, scale):
    """DIT:64-65."""
    return x * (1 + scale) + shift


# ------------------------------------------------------------------ attention
ATTENTION_Q_CHUNK = None   # full-size fp32 runs (tests/fullsize_parity.py): query rows per chunk of attention_chunked


def attention_chunked(q, k, v, num_heads: int, chunk: int) -> torch.Tensor:
    """The formula of DIT:28-61 evaluated in query-row chunks with plain matmul + softmax in the tensors' own dtype: a row's
    softmax sees all keys at once, so nothing changes mathematically, but the [heads, S, S] score tensor (171 GB in fp32 at
    S = 32760) is never materialised.  Used for the fp32-math oracle at production size only."""
    b, sq, hd = q.shape
    d = hd // num_heads
    kh = k.reshape(b, k.shape[1], num_heads, d).permute(0, 2, 3, 1)          # [b, h, d, Skv]
    vh = v.reshape(b, v.shape[1], num_heads, d).transpose(1, 2)               # [b, h, Skv, d]
    out = torch.empty_like(q)
    for s0 in range(0, sq, chunk):
        qh = q[:, s0:s0 + chunk].reshape(b, -1, num_heads, d).transpose(1, 2)  # [b, h, c, d]
        p = torch.softmax((qh @ kh) * (1.0 / math.sqrt(d)), dim=-1)
        out[:, s0:s0 + chunk] = (p @ vh).transpose(1, 2).reshape(b, -1, hd)
    return out


def attention(q, k, v, num_heads: int) -> torch.Tensor:
    """DIT:28-61 (SDPA branch) — softmax(q k^T / sqrt(d)) v, no mask; [B,S,H*d] layout."""
    b, sq, hd = q.shape
    d = hd // num_heads
    if ATTENTION_Q_CHUNK and q.dtype in (torch.float32, torch.float64) and sq > ATTENTION_Q_CHUNK:
        return attention_chunked(q, k, v, num_heads, ATTENTION_Q_CHUNK)

    def split(t):
        return t.reshape(b, t.shape[1], num_heads, d).transpose(1, 2)

    o = F.scaled_dot_product_attention(split(q), split(k), split(v))
    return o.transpose(1, 2).reshape(b, sq, hd)


def attention_fp64(q, k, v, num_heads: int) -> torch.Tensor:
    """Independent full-tensor fp64 reference of the same formula (no SDPA), for kernel tests."""
    b, sq, hd = q.shape
    d = hd // num_heads
    qd = q.double().reshape(b, sq, num_heads, d).transpose(1, 2)
    kd = k.double().reshape(b, k.shape[1], num_heads, d).transpose(1, 2)
    vd = v.double().reshape(b, v.shape[1], num_heads, d).transpose(1, 2)
    p = torch.softmax(qd @ kd.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (p @ vd).transpose(1, 2).reshape(b, sq, hd)


# ------------------------------------------------------------------ DiT block (DIT:124-230)
LINEAR = F.linear   # tests swap in oracle.fp8_oracle.fp8_linear to restate BASELINE config 5 (VRAM:115-151)


def _lin(x, sd, name):
    return LINEAR(x, sd[name + ".weight"], sd.get(name + ".bias"))


def self_attention(x, freqs, sd, pre, num_heads, eps):
    """DIT:140-147."""
    q = rms_norm(_lin(x, sd, pre + "q"), sd[pre + "norm_q.weight"], eps)
    k = rms_norm(_lin(x, sd, pre + "k"), sd[pre + "norm_k.weight"], eps)
    v = _lin(x, sd, pre + "v")
    q = rope_apply(q, freqs, num_heads)
    k = rope_apply(k, freqs, num_heads)
    return _lin(attention(q, k, v, num_heads), sd, pre + "o")


def cross_attention(x, ctx, sd, pre, num_heads, eps):
    """DIT:170-186, has_image_input=False branch: all context rows attended, no mask."""
    q = rms_norm(_lin(x, sd, pre + "q"), sd[pre + "norm_q.weight"], eps)
    k = rms_norm(_lin(ctx, sd, pre + "k"), sd[pre + "norm_k.weight"], eps)
    v = _lin(ctx, sd, pre + "v")
    return _lin(attention(q, k, v, num_heads), sd, pre + "o")


def dit_block(x, context, t_mod, freqs, sd, pre, num_heads, eps=1e-6):
    """DIT:214-230.  x [B,S,D], context [B,L,D], t_mod [B,6,D], freqs [S,d/2] complex128."""
    mod = sd[pre + "modulation"].to(dtype=t_mod.dtype) + t_mod
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=1)
    h = modulate(layer_norm(x, eps=eps), shift_msa, scale_msa)
    x = x + gate_msa * self_attention(h, freqs, sd, pre + "self_attn.", num_heads, eps)
    h = layer_norm(x, sd[pre + "norm3.weight"], sd[pre + "norm3.bias"], eps)
    x = x + cross_attention(h, context, sd, pre + "cross_attn.", num_heads, eps)
    h = modulate(layer_norm(x, eps=eps), shift_mlp, scale_mlp)
    h = _lin(F.gelu(_lin(h, sd, pre + "ffn.0"), approximate="tanh"), sd, pre + "ffn.2")
    return x + gate_mlp * h


# ------------------------------------------------------------------ model_fn (GF:1349-1591)
def patch_embed(x, weight, bias):
    """DIT:341-349 / GF:85-94 — Conv3d k=s=(1,2,2) then 'b c f h w -> b (f h w) c'."""
    y = F.conv3d(x, weight, bias, stride=weight.shape[2:])
    grid = tuple(y.shape[2:])
    return y.flatten(2).transpose(1, 2).contiguous(), grid


def unpatchify(x, grid, out_dim, patch=(1, 2, 2)):
    """DIT:351-356 — 'b (f h w) (x y z c) -> b c (f x) (h y) (w z)'."""
    f, h, w = grid
    px, py, pz = patch
    b = x.shape[0]
    x = x.reshape(b, f, h, w, px, py, pz, out_dim)
    x = x.permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.reshape(b, out_dim, f * px, h * py, w * pz)


def head(x, t, sd, pre="head.", eps=1e-6):
    """DIT:262-269 — LN, (1+scale)*+shift with modulation[1,2,D] + t[:,None], Linear(D, 64)."""
    mod = sd[pre + "modulation"].to(dtype=t.dtype) + t.unsqueeze(1)
    shift, scale = mod.chunk(2, dim=1)
    return _lin(layer_norm(x, eps=eps) * (1 + scale) + shift, sd, pre + "head")


def time_embed(timestep, sd, freq_dim, dim):
    """DIT:314-320 + GF:1441-1442."""
    e = sinusoidal_embedding_1d(freq_dim, timestep)
    t = _lin(F.silu(_lin(e, sd, "time_embedding.0")), sd, "time_embedding.2")
    t_mod = _lin(F.silu(t), sd, "time_projection.1").unflatten(1, (6, dim))
    return t, t_mod


def text_embed(context, sd):
    """DIT:309-313."""
    return _lin(F.gelu(_lin(context, sd, "text_embedding.0"), approximate="tanh"), sd, "text_embedding.2")


def model_fn(dit_sd, cfg, latents, timestep, context, y=None, controlnet_sd=None, control_latents=None,
             num_controlnet_layers=0, tap=None):
    """GF:1349-1591 for the Goal-Force inference configuration: no clip/vace/usp/teacache; ControlNet
    = patch-embed of the control latents, N DiT blocks on control tokens, zero-conv(state_i) added to
    x after DiT block i (non-strided path GF:1563-1570).  `tap(i, x)` (tests): called with the residual stream right after
    DiT block i, before the ControlNet injection."""
    dim, nh, eps = cfg["dim"], cfg["num_heads"], cfg["eps"]
    t, t_mod = time_embed(timestep, dit_sd, cfg["freq_dim"], dim)
    ctx = text_embed(context, dit_sd)
    x = latents if y is None else torch.cat([latents, y], dim=1)
    x, (f, h, w) = patch_embed(x, dit_sd["patch_embedding.weight"], dit_sd["patch_embedding.bias"])
    freqs = rope_freqs_3d(dim // nh, f, h, w).to(latents.device)
    states = []
    if controlnet_sd is not None:
        c, _ = patch_embed(control_latents,
                           controlnet_sd["controlnet_patch_embedding.patch_embedding.weight"],
                           controlnet_sd["controlnet_patch_embedding.patch_embedding.bias"])
        for i in range(num_controlnet_layers):
            c = dit_block(c, ctx, t_mod, freqs, controlnet_sd, f"controlnet_dit.blocks.{i}.", nh, eps)
            states.append(c)
    for i in range(cfg["num_layers"]):
        x = dit_block(x, ctx, t_mod, freqs, dit_sd, f"blocks.{i}.", nh, eps)
        if tap is not None:
            tap(i, x)
        if controlnet_sd is not None and i < num_controlnet_layers:
            zw = controlnet_sd[f"controlnet_zero_convs_after.{i}.weight"]
            zb = controlnet_sd[f"controlnet_zero_convs_after.{i}.bias"]
            x = x + F.conv1d(states[i].transpose(1, 2), zw, zb).transpose(1, 2)
    x = head(x, t, dit_sd, eps=eps)
    return unpatchify(x, (f, h, w), cfg["out_dim"])


# ------------------------------------------------------------------ scheduler + loop (FM, GF:697-723)
def flow_match_sigmas(num_inference_steps=50, shift=5.0, sigma_min=0.0, sigma_max=1.0, denoising_strength=1.0,
                      extra_one_step=True, num_train_timesteps=1000):
    """FM:34-60 — returns (sigmas, timesteps) fp32."""
    start = sigma_min + (sigma_max - sigma_min) * denoising_strength
    if extra_one_step:
        s = torch.linspace(start, sigma_min, num_inference_steps + 1)[:-1]
    else:
        s = torch.linspace(start, sigma_min, num_inference_steps)
    s = shift * s / (1 + (shift - 1) * s)
    return s, s * num_train_timesteps


def euler_step(model_output, step_id, sample, sigmas):
    """FM:72-82 with timestep == timesteps[step_id]."""
    sigma = sigmas[step_id]
    sigma_next = 0 if step_id + 1 >= len(sigmas) else sigmas[step_id + 1]
    return sample + model_output * (sigma_next - sigma)


def cfg_combine(posi, nega, cfg_scale):
    """GF:716."""
    return nega + cfg_scale * (posi - nega)


def denoise_loop(experts, latents, ctx_posi, ctx_nega, y, control_latents, num_inference_steps, cfg_scale=5.0,
                 shift=5.0, boundary=0.875, dtype=torch.bfloat16):
    """GF:697-723.  experts = [(dit_sd, cfg, controlnet_sd, n_layers), (dit2_sd, cfg, controlnet2_sd, n_layers)]."""
    sigmas, timesteps = flow_match_sigmas(num_inference_steps, shift)
    cur = 0
    for i, ts in enumerate(timesteps):
        if ts.item() < boundary * 1000 and cur == 0 and len(experts) > 1:
            cur = 1
        sd, cfg, csd, nl = experts[cur]
        t = ts.unsqueeze(0).to(dtype)
        posi = model_fn(sd, cfg, latents, t, ctx_posi, y, csd, control_latents, nl)
        if cfg_scale != 1.0:
            nega = model_fn(sd, cfg, latents, t, ctx_nega, y, csd, control_latents, nl)
            pred = cfg_combine(posi, nega, cfg_scale)
        else:
            pred = posi
        latents = euler_step(pred, i, latents, sigmas)
    return latents


# ------------------------------------------------------------------ random weights for tests / bench
def block_param_shapes(dim, ffn_dim):
    shapes = {}
    for att in ("self_attn", "cross_attn"):
        for p in ("q", "k", "v", "o"):
            shapes[f"{att}.{p}.weight"] = (dim, dim)
            shapes[f"{att}.{p}.bias"] = (dim,)
        shapes[f"{att}.norm_q.weight"] = (dim,)
        shapes[f"{att}.norm_k.weight"] = (dim,)
    shapes["norm3.weight"] = (dim,)
    shapes["norm3.bias"] = (dim,)
    shapes["ffn.0.weight"] = (ffn_dim, dim)
    shapes["ffn.0.bias"] = (ffn_dim,)
    shapes["ffn.2.weight"] = (dim, ffn_dim)
    shapes["ffn.2.bias"] = (dim,)
    shapes["modulation"] = (1, 6, dim)
    return shapes


def random_block_sd(dim, ffn_dim, prefix="", seed=0, dtype=torch.bfloat16, std=None):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shp in block_param_shapes(dim, ffn_dim).items():
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name == "norm3.weight":
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name == "modulation":
            t = torch.randn(shp, generator=g) / dim ** 0.5
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        else:
            s = std if std is not None else 1.0 / math.sqrt(shp[1])
            t = s * torch.randn(shp, generator=g)
        sd[prefix + name] = t.to(dtype)
    return sd
