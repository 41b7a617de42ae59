"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Functional torch-CPU restatement of the Wan VAE decode path
(diffsynth/models/wan_video_vae.py "VAE": VideoVAE_.decode 1011-1034, Decoder3d 736-838, ResidualBlock
267-301, AttentionBlock 304-342, Resample 82-174, CausalConv3d 33-52, RMS_norm 55-70,
WanVideoVAE.tiled_decode / build_mask 1081-1152).  Pinned against tests/golden/g6_vae.npz, produced by the
reference's own WanVideoVAE with seeded random weights (tests/test_vae.py).

Formulation: every causal conv keeps a 2-frame history that starts as zeros and becomes the last two frames of
(history ++ input) after each chunk — equivalent to the reference's feat_cache/None/'Rep' bookkeeping.
Tensors are NCTHW like the reference; dtype AND device follow the inputs (bf16 = the reference's own arithmetic; on a GPU the same
torch ops run on torch-ROCm's kernels: tests/fullsize_vae_parity.py uses that for the full-size decode / encode)."""
import math

import torch
import torch.nn.functional as F

MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
        0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
       3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class _State:
    def __init__(self):
        self.hist = {}


def _causal_conv(x, sd, name, st):
    w, b = sd[name + ".weight"], sd[name + ".bias"]
    kt, kh = w.shape[2], w.shape[3]
    if kt > 1:
        h = st.hist.get(name)
        if h is None:
            h = torch.zeros_like(x[:, :, :1]).repeat(1, 1, 2, 1, 1)
        xin = torch.cat([h, x], dim=2)
        st.hist[name] = xin[:, :, -2:].clone()
    else:
        xin = x
    p = kh // 2
    return F.conv3d(F.pad(xin, (p, p, p, p, 0, 0)), w, b)


def _rms(x, gamma, dim=1):
    return F.normalize(x, dim=dim) * (x.shape[dim] ** 0.5) * gamma


def _res(x, sd, name, st):
    h = _causal_conv(x, sd, name + ".shortcut", st) if (name + ".shortcut.weight") in sd else x
    y = F.silu(_rms(x, sd[name + ".residual.0.gamma"]))
    y = _causal_conv(y, sd, name + ".residual.2", st)
    y = F.silu(_rms(y, sd[name + ".residual.3.gamma"]))
    y = _causal_conv(y, sd, name + ".residual.6", st)
    return y + h


def _attn(x, sd, name):
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = _rms(y, sd[name + ".norm.gamma"])
    qkv = F.conv2d(y, sd[name + ".to_qkv.weight"], sd[name + ".to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, 1, 3 * c, h * w).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    o = F.conv2d(o, sd[name + ".proj.weight"], sd[name + ".proj.bias"])
    return o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def _up(x, sd, name, st, first):
    b, c, t, h, w = x.shape
    if (name + ".time_conv.weight") in sd and not first:
        y = _causal_conv(x, sd, name + ".time_conv", st)          # [b, 2c, t, h, w]
        y = y.reshape(b, 2, c, t, h, w)
        x = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, 2 * t, h, w)
        t = 2 * t
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(y)
    y = F.conv2d(y, sd[name + ".resample.1.weight"], sd[name + ".resample.1.bias"], padding=1)
    return y.reshape(b, t, -1, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)


def _decoder_chunk(x, sd, st, first):
    x = _causal_conv(x, sd, "decoder.conv1", st)
    x = _res(x, sd, "decoder.middle.0", st)
    x = _attn(x, sd, "decoder.middle.1")
    x = _res(x, sd, "decoder.middle.2", st)
    i = 0
    while True:
        n = f"decoder.upsamples.{i}"
        if (n + ".residual.0.gamma") in sd:
            x = _res(x, sd, n, st)
        elif (n + ".resample.1.weight") in sd:
            x = _up(x, sd, n, st, first)
        else:
            break
        i += 1
    x = F.silu(_rms(x, sd["decoder.head.0.gamma"]))
    return _causal_conv(x, sd, "decoder.head.2", st)


def decode(z, sd):
    """VideoVAE_.decode: z [1,16,T,h,w] -> [1,3,4T-3,8h,8w] (un-clamped).  sd keys without the 'model.' prefix."""
    mean = torch.tensor(MEAN).to(z).view(1, -1, 1, 1, 1)                     # .to(z): dtype and device of the input
    inv_std = (1.0 / torch.tensor(STD)).to(z).view(1, -1, 1, 1, 1)
    z = z / inv_std + mean
    x = F.conv3d(z, sd["conv2.weight"], sd["conv2.bias"])
    st = _State()
    outs = [_decoder_chunk(x[:, :, i:i + 1], sd, st, first=(i == 0)) for i in range(x.shape[2])]
    return torch.cat(outs, dim=2)


def _down(x, sd, name, st):
    """Resample downsample2d/3d (VAE:101-112, 159-174)."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.conv2d(F.pad(y, (0, 1, 0, 1)), sd[name + ".resample.1.weight"], sd[name + ".resample.1.bias"], stride=2)
    x = y.reshape(b, t, c, h // 2, w // 2).permute(0, 2, 1, 3, 4)
    if (name + ".time_conv.weight") in sd:
        prev = st.hist.get(name + ".time_conv")
        st.hist[name + ".time_conv"] = x[:, :, -1:].clone()
        if prev is not None:
            x = F.conv3d(torch.cat([prev, x], 2), sd[name + ".time_conv.weight"], sd[name + ".time_conv.bias"],
                         stride=(2, 1, 1))
    return x


def _encoder_chunk(x, sd, st):
    x = _causal_conv(x, sd, "encoder.conv1", st)
    i = 0
    while True:
        n = f"encoder.downsamples.{i}"
        if (n + ".residual.0.gamma") in sd:
            x = _res(x, sd, n, st)
        elif (n + ".resample.1.weight") in sd:
            x = _down(x, sd, n, st)
        else:
            break
        i += 1
    x = _res(x, sd, "encoder.middle.0", st)
    x = _attn(x, sd, "encoder.middle.1")
    x = _res(x, sd, "encoder.middle.2", st)
    x = F.silu(_rms(x, sd["encoder.head.0.gamma"]))
    return _causal_conv(x, sd, "encoder.head.2", st)


def encode(video, sd):
    """VideoVAE_.encode (VAE:988-1010): video [1,3,T,H,W] in [-1,1] -> normalised mu [1,16,1+(T-1)//4,H/8,W/8]."""
    st = _State()
    t = video.shape[2]
    outs = []
    for i in range(1 + (t - 1) // 4):
        chunk = video[:, :, :1] if i == 0 else video[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(_encoder_chunk(chunk, sd, st))
    out = torch.cat(outs, 2)
    mu, _ = F.conv3d(out, sd["conv1.weight"], sd["conv1.bias"]).chunk(2, dim=1)
    mean = torch.tensor(MEAN).to(mu).view(1, -1, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(STD)).to(mu).view(1, -1, 1, 1, 1)
    return (mu - mean) * inv_std


def tiled_encode(video, sd, tile_size, tile_stride, up=8):
    """WanVideoVAE.tiled_encode (VAE:1155-1203); sizes in pixels."""
    _, _, T, H, W = video.shape
    (sh, sw), (th, tw) = tile_size, tile_stride
    tasks = []
    for h in range(0, H, th):
        if h - th >= 0 and h - th + sh >= H:
            continue
        for w in range(0, W, tw):
            if w - tw >= 0 and w - tw + sw >= W:
                continue
            tasks.append((h, h + sh, w, w + sw))
    oT = (T + 3) // 4
    weight = torch.zeros((1, 1, oT, H // up, W // up), dtype=video.dtype, device=video.device)
    values = torch.zeros((1, 16, oT, H // up, W // up), dtype=video.dtype, device=video.device)
    for h, h_, w, w_ in tasks:
        tile = encode(video[:, :, :, h:h_, w:w_], sd)
        mh = _ramp(tile.shape[3], h == 0, h_ >= H, (sh - th) // up)
        mw = _ramp(tile.shape[4], w == 0, w_ >= W, (sw - tw) // up)
        mask = torch.minimum(mh[:, None].expand(-1, tile.shape[4]), mw[None, :].expand(tile.shape[3], -1))
        mask = mask.view(1, 1, 1, *mask.shape).to(video)
        values[:, :, :, h // up:h // up + tile.shape[3], w // up:w // up + tile.shape[4]] += tile * mask
        weight[:, :, :, h // up:h // up + tile.shape[3], w // up:w // up + tile.shape[4]] += mask
    return values / weight


def _ramp(length, left_bound, right_bound, border):
    x = torch.ones((length,))
    if not left_bound:
        x[:border] = (torch.arange(border) + 1) / border
    if not right_bound:
        x[-border:] = torch.flip((torch.arange(border) + 1) / border, dims=(0,))
    return x


def tiled_decode(z, sd, tile_size, tile_stride, up=8):
    """WanVideoVAE.tiled_decode (VAE:1103-1152), accumulators in z.dtype."""
    _, _, T, H, W = z.shape
    (sh, sw), (th, tw) = tile_size, tile_stride
    tasks = []
    for h in range(0, H, th):
        if h - th >= 0 and h - th + sh >= H:
            continue
        for w in range(0, W, tw):
            if w - tw >= 0 and w - tw + sw >= W:
                continue
            tasks.append((h, h + sh, w, w + sw))
    oT = 4 * T - 3
    weight = torch.zeros((1, 1, oT, H * up, W * up), dtype=z.dtype, device=z.device)
    values = torch.zeros((1, 3, oT, H * up, W * up), dtype=z.dtype, device=z.device)
    for h, h_, w, w_ in tasks:
        tile = decode(z[:, :, :, h:h_, w:w_], sd)
        mh = _ramp(tile.shape[3], h == 0, h_ >= H, (sh - th) * up)
        mw = _ramp(tile.shape[4], w == 0, w_ >= W, (sw - tw) * up)
        mask = torch.minimum(mh[:, None].expand(-1, tile.shape[4]), mw[None, :].expand(tile.shape[3], -1))
        mask = mask.view(1, 1, 1, *mask.shape).to(z)
        values[:, :, :, h * up:h * up + tile.shape[3], w * up:w * up + tile.shape[4]] += tile * mask
        weight[:, :, :, h * up:h * up + tile.shape[3], w * up:w * up + tile.shape[4]] += mask
    return (values / weight).clamp_(-1, 1)
