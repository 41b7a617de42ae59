"""VAE decoder: (CPU) oracle pinned to the reference's own WanVideoVAE outputs; (GPU) HIP decode path vs the
same goldens.  Tolerances: bf16 oracle vs reference bf16 bit-exact (same torch ops); HIP vs fp32 golden
rel-L2 <= max(1.5e-2, 1.5x the reference-bf16's own distance from fp32) — the decoder is ~50 bf16 convs deep."""
import math
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN, rel_l2
from oracle import vae_oracle as vo

BF = torch.bfloat16
torch.set_grad_enabled(False)


def _fixture():
    g = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    sd = gi.vae_decoder_sd(list(g["names"]), g["shapes"], seed=61)
    gen = torch.Generator().manual_seed(62)
    z1 = torch.randn((1, 16, 3, 8, 8), generator=gen).to(BF)
    z2 = torch.randn((1, 16, 2, 12, 16), generator=gen).to(BF)
    v1 = (torch.rand((1, 3, 9, 64, 64), generator=gen) * 2 - 1).to(BF)
    v2 = (torch.rand((1, 3, 5, 96, 128), generator=gen) * 2 - 1).to(BF)
    assert gi.same_checksum(gi.checksum(sd), g["ck_weights"]) and gi.same_checksum(gi.checksum([z1, z2, v1, v2]), g["ck_inputs"])
    _fixture.videos = (v1, v2)
    return g, sd, z1, z2


def test_decoder_layout_matches_reference_state_dict():
    from goal_force_amd.vae import WanVideoVAE, decoder_layout
    g = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    from goal_force_amd.vae import encoder_layout
    shapes = {**encoder_layout()[1], **decoder_layout()[1]}
    ref = {n: tuple(int(v) for v in s if int(v) > 0) for n, s in zip(g["names"], g["shapes"])}
    assert ref == {k: tuple(v) for k, v in shapes.items()}
    assert sorted(WanVideoVAE().state_dict().keys()) == sorted("model." + n for n in ref)
    assert sum(int(np.prod(s)) for s in ref.values()) == 126_892_531   # every parameter of the reference VAE


def test_oracle_decode_matches_reference():
    g, sd, z1, z2 = _fixture()
    raw = vo.decode(z1, sd)
    assert torch.equal(raw, gi.from_u16(g["decode_raw_bf16"]))               # bit-exact in the reference's arithmetic
    assert torch.equal(raw.clamp(-1, 1), gi.from_u16(g["decode_bf16"]))
    sd32 = {k: v.float() for k, v in sd.items()}
    assert rel_l2(vo.decode(z1.float(), sd32), torch.from_numpy(g["decode_raw_f32"])) < 1e-5


def test_oracle_tiled_decode_matches_reference():
    g, sd, z1, z2 = _fixture()
    got = vo.tiled_decode(z2, sd, (8, 8), (4, 4))
    assert torch.equal(got, gi.from_u16(g["tiled_bf16"]))
    sd32 = {k: v.float() for k, v in sd.items()}
    assert rel_l2(vo.tiled_decode(z2.float(), sd32, (8, 8), (4, 4)), torch.from_numpy(g["tiled_f32"])) < 1e-5


def test_oracle_encode_matches_reference():
    g, sd, z1, z2 = _fixture()
    v1, v2 = _fixture.videos
    assert torch.equal(vo.encode(v1, sd), gi.from_u16(g["encode_bf16"]))
    assert torch.equal(vo.tiled_encode(v2, sd, (64, 64), (32, 32)), gi.from_u16(g["encode_tiled_bf16"]))
    sd32 = {k: v.float() for k, v in sd.items()}
    assert rel_l2(vo.encode(v1.float(), sd32), torch.from_numpy(g["encode_f32"])) < 1e-5


@pytest.fixture(autouse=True)
def _unrotated_yardstick_gemm():
    """The convolution is compared BIT FOR BIT with `gf_vae_im2col + gf_gemm_bf16`: pin that yardstick GEMM to the K order
    the implicit-GEMM convolution uses (k tiles 0, 1, 2, ...).  The 4-wave GEMM kernel that large dense shapes go through by
    default starts each column tile's K loop at a staggered tile (same sum, rotated order: test_kernels_gpu.py)."""
    from goal_force_amd import ops
    with ops.options(a4_stagger=0):
        yield


def _gpu_vae(sd):
    from goal_force_amd.vae import WanVideoVAE
    v = WanVideoVAE()
    v.load_state_dict({"model." + k: t for k, t in sd.items()}, strict=True)
    return v.to(BF).cuda()


@pytest.mark.gpu
def test_hip_vae_im2col_and_norm_kernels_exact():
    from goal_force_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    T, H, W, C = 2, 5, 6, 16
    x = torch.randn((T, H, W, C), generator=g).to(BF)
    cache = torch.randn((2, H, W, C), generator=g).to(BF)
    cols = ops.vae_im2col(x.cuda(), cache.cuda(), 3, 3, 448).cpu()
    seq = torch.cat([cache, x], 0).permute(3, 0, 1, 2)[None].float()            # [1,C,T+2,H,W]
    pad = F.pad(seq, (1, 1, 1, 1, 0, 0))
    patches = pad.unfold(2, 3, 1).unfold(3, 3, 1).unfold(4, 3, 1)                # [1,C,T,H,W,3,3,3]
    ref = patches[0].permute(1, 2, 3, 4, 5, 6, 0).reshape(T * H * W, 27 * C).to(BF)
    assert torch.equal(cols[:, :432], ref) and float(cols[:, 432:].abs().sum()) == 0
    # upsample-folded 3x3 gather == unfold of the nearest-exact upsampled image
    cols2 = ops.vae_im2col(x.cuda(), None, 1, 3, 192, upsample2x=True).cpu()
    up = F.interpolate(x.permute(0, 3, 1, 2).float(), scale_factor=(2.0, 2.0), mode="nearest-exact")
    p2 = F.pad(up, (1, 1, 1, 1)).unfold(2, 3, 1).unfold(3, 3, 1)                 # [T,C,2H,2W,3,3]
    ref2 = p2.permute(0, 2, 3, 4, 5, 1).reshape(T * 4 * H * W, 9 * C).to(BF)
    assert torch.equal(cols2[:, :144], ref2)
    # RMS_norm + SiLU against the oracle's eager bf16 ops
    xx = (torch.randn((7, 9, 96), generator=g) * 3).to(BF)
    gam = (1 + 0.1 * torch.randn(96, generator=g)).to(BF)
    ref3 = F.silu(F.normalize(xx, dim=-1) * (96 ** 0.5) * gam)
    got3 = ops.vae_rmsnorm_silu(xx.cuda(), gam.cuda(), silu=True).cpu()
    bad = ((got3.view(torch.int16).int() - ref3.view(torch.int16).int()).abs() > 1).float().mean()
    assert rel_l2(got3.float(), ref3.float()) < 3e-3 and float(bad) < 5e-3
    # softmax rows + transpose
    s = torch.randn((20, 24), generator=g).to(BF)
    p = ops.softmax_rows(s.cuda(), 0.3, 64).cpu()
    assert rel_l2(p[:, :24].float(), torch.softmax(s.float() * 0.3, -1)) < 3e-3 and float(p[:, 24:].abs().sum()) == 0
    tp = ops.transpose_pad(s.cuda(), 64).cpu()
    assert torch.equal(tp[:, :20], s.t()) and float(tp[:, 20:].abs().sum()) == 0
    for R, C, rpad, ld in ((333, 136, 384, 200), (64, 64, 64, 64), (1000, 5120, 1024, 5120), (7, 12, 8, 12)):   # 64 x 64 vector tiles / the 2-byte kernel
        big = torch.randn((R, ld), generator=g).to(BF).cuda()
        tq = ops.transpose_pad(big[:, :C], rpad).cpu()
        assert torch.equal(tq[:, :R], big[:, :C].cpu().t()) and float(tq[:, R:].abs().sum()) == 0, (R, C, rpad)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (T, H, W, C, N, kt, ks, gather kwargs, residual)
    dict(T=2, H=5, W=6, C=16, N=24, kt=3, ks=3),                                   # causal 3x3x3, K 432 -> 448
    dict(T=4, H=18, W=20, C=96, N=96, kt=3, ks=3, resid=True),                     # decoder top level (C=96), > 1 M tile
    dict(T=1, H=9, W=7, C=192, N=96, kt=1, ks=3, gather=dict(upsample2x=True)),    # Resample upsample2d conv
    dict(T=3, H=8, W=10, C=24, N=24, kt=1, ks=3, gather=dict(downsample2=True)),   # Resample downsample2d conv
    dict(T=4, H=6, W=5, C=32, N=32, kt=3, ks=1, gather=dict(t_stride=2, t_off=1, t_out=2)),   # encoder time_conv
    dict(T=2, H=6, W=5, C=32, N=64, kt=3, ks=1, gather=dict(t_off=1, t_out=1)),    # decoder time_conv, one frame at a time
    dict(T=1, H=12, W=12, C=8, N=96, kt=3, ks=3),                                  # encoder conv1 on RGB padded to 8
    dict(T=1, H=40, W=40, C=384, N=384, kt=3, ks=3, resid=True),                   # widest level: K 10368, 2 N tiles
    dict(T=2, H=60, W=104, C=96, N=96, kt=3, ks=3),                                # 49 M tiles: the XCD-aware tile map
    dict(T=3, H=33, W=20, C=96, N=8, kt=3, ks=3),                                  # RGB head (3 -> 8 columns), ragged last M tile
    dict(T=1, H=16, W=16, C=64, N=128, kt=3, ks=3, resid=True),                    # exactly one 128-column tile, K 1728
    dict(T=1, H=16, W=16, C=64, N=136, kt=1, ks=3),                                # just past it: the 256-wide tile again
    dict(T=1, H=16, W=16, C=8, N=16, kt=1, ks=1),                                  # a single K tile (nk = 1)
    dict(T=2, H=30, W=52, C=192, N=96, kt=1, ks=3, gather=dict(upsample2x=True), resid=True),   # upsample conv, 49 M tiles
])
def test_hip_implicit_conv_bit_identical_to_im2col_gemm(case):
    """gf_conv3d_bf16 (no patch matrix) == gf_vae_im2col + gf_gemm_bf16, bit for bit, in every gather mode."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(11)
    T, H, W, C, N, kt, ks = (case[k] for k in ("T", "H", "W", "C", "N", "kt", "ks"))
    gather = case.get("gather", {})
    k = kt * ks * ks * C
    kpad = -(-k // 64) * 64
    x = torch.randn((T, H, W, C), generator=g).to(BF).cuda()
    cache = torch.randn((2, H, W, C), generator=g).to(BF).cuda() if kt == 3 else None
    w = torch.zeros((N, kpad), dtype=BF)
    w[:, :k] = (torch.randn((N, k), generator=g) / k ** 0.5).to(BF)
    w = w.cuda()
    b = torch.randn((N,), generator=g).to(BF).cuda()
    cols = ops.vae_im2col(x, cache, kt, ks, kpad, **gather)
    resid = torch.randn((cols.shape[0], N), generator=g).to(BF).cuda() if case.get("resid") else None
    ref = ops.gemm(cols, w, b, epilogue=ops.EPI_BIAS if resid is None else ops.EPI_BIAS_RESID, resid=resid)
    got = ops.vae_conv3d(x, cache, w, b, kt, ks, resid=resid, **gather)
    assert got.shape == ref.shape and torch.equal(got, ref)
    assert float(ref.float().abs().max()) > 0.5                                    # not comparing zeros


@pytest.mark.gpu
@pytest.mark.parametrize("T,H,W,C,N,ks", [(3, 9, 11, 96, 96, 3), (5, 18, 20, 96, 8, 3), (2, 6, 5, 32, 64, 1), (4, 13, 7, 192, 192, 3),
                                          (1, 5, 5, 16, 24, 3), (9, 20, 70, 96, 96, 3), (22, 13, 33, 96, 96, 3), (1, 6, 32, 96, 96, 3),
                                          (7, 30, 45, 96, 8, 3), (3, 9, 40, 96, 16, 3), (13, 8, 32, 96, 8, 3)])
def test_hip_conv_history_in_front_bit_identical(T, H, W, C, N, ks):
    """The pointer-per-row gather (history frames in front of src, one buffer) == the general gather with a separate cache
    == gf_vae_im2col + gf_gemm_bf16, bit for bit, with and without a residual."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(T * 100 + H)
    k = 3 * ks * ks * C
    kpad = -(-k // 64) * 64
    buf = torch.randn((T + 2, H, W, C), generator=g).to(BF).cuda()
    x, cache = buf[2:], buf[:2].clone()
    w = torch.zeros((N, kpad), dtype=BF)
    w[:, :k] = (torch.randn((N, k), generator=g) / k ** 0.5).to(BF)
    w = w.cuda()
    b = torch.randn((N,), generator=g).to(BF).cuda()
    resid = torch.randn((T * H * W, N), generator=g).to(BF).cuda()
    ref = ops.gemm(ops.vae_im2col(x.contiguous(), cache, 3, ks, kpad), w, b)
    assert torch.equal(ops.vae_conv3d(x, None, w, b, 3, ks, history_in_front=True), ref)
    assert torch.equal(ops.vae_conv3d(x.contiguous(), cache, w, b, 3, ks), ref)
    ref_r = ops.gemm(ops.vae_im2col(x.contiguous(), cache, 3, ks, kpad), w, b, epilogue=ops.EPI_BIAS_RESID, resid=resid)
    assert torch.equal(ops.vae_conv3d(x, None, w, b, 3, ks, resid=resid, history_in_front=True), ref_r)
    from goal_force_amd._lib import GoalForceError
    with pytest.raises(GoalForceError):
        ops.vae_conv3d(x.clone(), None, w, b, 3, ks, history_in_front=True)        # no room in front of a fresh tensor


@pytest.mark.gpu
def test_hip_direct_conv_c96_equals_implicit_gemm_at_tile_size():
    """The 96-channel level of a production tile's first frames (240 x 416; C = N = 96, 3x3x3, history in front): the direct
    convolution (gf_conv_direct.hip: 6 x 32 pixel patches walking the frames, input halo staged once per frame, three
    accumulator sets) against the implicit GEMM it replaces (options(conv_direct=0)) — bit for bit, with and without the residual, over
    several frame segments."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(96)
    T, H, W, C = 7, 240, 416, 96
    k = 27 * C
    kpad = -(-k // 64) * 64
    buf = (torch.randn((T + 2, H, W, C), generator=g) * 0.7).to(BF).cuda()
    x = buf[2:]
    w = torch.zeros((C, kpad), dtype=BF)
    w[:, :k] = (torch.randn((C, k), generator=g) / k ** 0.5).to(BF)
    w = w.cuda()
    b = torch.randn((C,), generator=g).to(BF).cuda()
    resid = torch.randn((T * H * W, C), generator=g).to(BF).cuda()
    with ops.options(conv_direct=0):
        ref = ops.vae_conv3d(x, None, w, b, 3, 3, history_in_front=True)
        ref_r = ops.vae_conv3d(x, None, w, b, 3, 3, resid=resid, history_in_front=True)
    got = ops.vae_conv3d(x, None, w, b, 3, 3, history_in_front=True)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {ref.numel()} differ"
    assert torch.equal(ops.vae_conv3d(x, None, w, b, 3, 3, resid=resid, history_in_front=True), ref_r)
    assert float(ref.float().abs().max()) > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("T,H,W,C,N", [(1, 4, 6, 192, 192), (3, 10, 14, 192, 384), (2, 30, 52, 384, 384), (5, 17, 23, 384, 384),
                                       (3, 60, 104, 192, 192), (2, 120, 208, 192, 192), (2, 6, 10, 192, 96), (1, 5, 7, 384, 200)])
def test_hip_padded_layout_conv_equals_implicit_gemm(T, H, W, C, N):
    """The 3x3x3 causal convolutions of the 192- / 384-channel levels on the padded-layout kernel (gf_conv_a4.hip: the activation in a
    zero-bordered buffer, taps as constant row shifts, the 4-wave GEMM loop on a 256 x 192 tile) against the implicit GEMM
    (gf_conv3d_bf16) on the same values — bit for bit, with a non-zero two-frame history, with and without the residual; sizes
    from one ragged row tile to a production tile's frames (120 x 208: 26 k padded rows per frame, rows of the border dropped);
    Cout = 96 and 200: a partly filled and a ragged second column tile (not shapes of the VAE; the kernel's edges)."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(T * 100 + H + C + N)
    k = 27 * C
    hist_and_x = (torch.randn((T + 2, H, W, C), generator=g) * 0.7).to(BF).cuda()
    w = (torch.randn((N, k), generator=g) / k ** 0.5).to(BF).cuda()
    b = torch.randn((N,), generator=g).to(BF).cuda()
    resid = torch.randn((T * H * W, N), generator=g).to(BF).cuda()
    with ops.options(conv_direct=0):
        ref = ops.vae_conv3d(hist_and_x[2:], None, w, b, 3, 3, history_in_front=True)
        ref_r = ops.vae_conv3d(hist_and_x[2:], None, w, b, 3, 3, resid=resid, history_in_front=True)
    buf, hist, cur = ops.padded_activation(T, H, W, C, "cuda")
    hist.copy_(hist_and_x[:2])
    cur.copy_(hist_and_x[2:])
    got = ops.vae_conv3d_padded(buf, w, b)
    assert got.shape == ref.shape == (T * H * W, N)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {ref.numel()} differ"
    assert torch.equal(ops.vae_conv3d_padded(buf, w, b, resid=resid), ref_r)
    assert float(ref.float().abs().max()) > 0.5
    # the borders are the zero padding: a buffer whose border was scribbled on gives other values (the test would be vacuous otherwise)
    if H * W <= 400:
        buf[:, 0].fill_(1.0)
        assert not torch.equal(ops.vae_conv3d_padded(buf, w, b), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("T,Hs,Ws", [(1, 3, 5), (3, 12, 16), (2, 60, 104)])
def test_hip_padded_layout_upsample_conv_equals_implicit_gemm(T, Hs, Ws):
    """The decoder's 384 -> 192 resample convolution behind the nearest-exact 2x upsample (VAE:82-96): the upsampled frames written
    once into the zero-bordered layout (gf_vae_upsample2x_padded) + gf_conv3d_padded_bf16 with kt = 1, against the implicit GEMM
    that folds the upsample into its gather — bit for bit."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(T * 1000 + Hs)
    C, N = 384, 192
    k = 9 * C
    x = (torch.randn((T, Hs, Ws, C), generator=g) * 0.7).to(BF).cuda()
    w = (torch.randn((N, k), generator=g) / k ** 0.5).to(BF).cuda()
    b = torch.randn((N,), generator=g).to(BF).cuda()
    ref = ops.vae_conv3d(x, None, w, b, 1, 3, upsample2x=True)
    buf, none, interior = ops.padded_activation(T, 2 * Hs, 2 * Ws, C, "cuda", history=False)
    assert none is None
    up = ops.vae_upsample2x_padded(x, buf)
    assert torch.equal(up, x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)) and up.data_ptr() == interior.data_ptr()
    assert float(buf[:, 0].abs().sum()) == 0 and float(buf[:, :, -1].abs().sum()) == 0
    got = ops.vae_conv3d_padded(buf, w, b, kt=1)
    assert got.shape == ref.shape == (T * 4 * Hs * Ws, N)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {ref.numel()} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("C", [192, 384])
def test_hip_rmsnorm_silu_into_the_padded_layout(C):
    """gf_vae_rmsnorm_silu_padded writes exactly gf_vae_rmsnorm_silu's values into the interior of the zero-bordered buffer and
    leaves the border and the history frames alone."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(C + 1)
    T, H, W = 3, 9, 13
    x = (torch.randn((T, H, W, C), generator=g) * 2.5).to(BF).cuda()
    gam = (1 + 0.1 * torch.randn(C, generator=g)).to(BF).cuda()
    buf, hist, cur = ops.padded_activation(T, H, W, C, "cuda")
    hist.fill_(3.0)
    for silu in (True, False):
        view = ops.vae_rmsnorm_silu_padded(x, gam, buf, silu=silu)
        assert torch.equal(view, ops.vae_rmsnorm_silu(x, gam, silu=silu)) and view.data_ptr() == cur.data_ptr()
        assert float(buf[:, 0].abs().sum()) == 0 and float(buf[:, -1].abs().sum()) == 0
        assert float(buf[:, :, 0].abs().sum()) == 0 and float(buf[:, :, -1].abs().sum()) == 0 and bool((hist == 3.0).all())


@pytest.mark.gpu
def test_hip_decoder_tile_same_bits_on_both_conv_paths():
    """One whole decoder tile (all levels, the feature caches across chunks) with the 192- / 384-channel convolutions on the
    padded-layout kernel and on the implicit GEMM (options(conv_padded=False)): the same frames, bit for bit; likewise the encoder."""
    from goal_force_amd import ops
    from goal_force_amd.vae import WanVideoVAE
    torch.manual_seed(11)
    vae = WanVideoVAE().to(BF).cuda()
    vae.frames_per_chunk = 2
    z = torch.randn((16, 5, 8, 12)).to(BF).cuda()
    got = vae.decode_tile_channels_last(z)
    with ops.options(conv_padded=False):
        ref = vae.decode_tile_channels_last(z)
    assert torch.equal(got, ref) and tuple(got.shape) == (17, 64, 96, 8)
    video = (torch.rand((3, 9, 64, 96)) * 2 - 1).to(BF).cuda()
    e = vae.encode_tile_channels_last(video)
    with ops.options(conv_padded=False):
        e_ref = vae.encode_tile_channels_last(video)
    assert torch.equal(e, e_ref)


@pytest.mark.gpu
def test_hip_padded_pool_is_released_by_every_entry_point_and_on_errors():
    """the zero-bordered conv inputs (~3 GB at a production tile) live for one decode / encode call — or one tile when
    the tile-level entry points are called directly — and are dropped when an exception passes through; the feature caches too.  A
    second stream inside one scope is refused (the buffers are shared by every convolution of the tile)."""
    from goal_force_amd._lib import GoalForceError
    from goal_force_amd.vae import WanVideoVAE
    torch.manual_seed(12)
    vae = WanVideoVAE().to(BF).cuda()
    z = torch.randn((16, 3, 8, 12)).to(BF).cuda()
    seen = []
    real = vae._decode_chunk

    def spy(P, x, first):
        out = real(P, x, first)
        seen.append(len(vae._pad_pool))
        return out
    vae._decode_chunk = spy
    ref = vae.decode_tile_channels_last(z)
    assert max(seen) > 0, "the tile went through the padded-layout kernel"
    assert vae._pad_pool == {} and vae._cache == {} and vae._pool_depth == 0
    got = vae.decode(z[None], tiled=False)
    assert vae._pad_pool == {} and vae._pool_depth == 0 and got.shape == (1, 3, 9, 64, 96)
    video = (torch.rand((3, 5, 64, 96)) * 2 - 1).to(BF).cuda()
    vae.encode_tile_channels_last(video)
    assert vae._pad_pool == {} and vae._cache == {}

    def boom(P, x, first):
        real(P, x, first)
        raise RuntimeError("boom")
    vae._decode_chunk = boom
    with pytest.raises(RuntimeError, match="boom"):
        vae.decode(z[None], tiled=True, tile_size=(8, 8), tile_stride=(4, 4))
    assert vae._pad_pool == {} and vae._cache == {} and vae._pool_depth == 0 and vae._pool_stream is None
    vae._decode_chunk = real
    assert torch.equal(vae.decode_tile_channels_last(z), ref), "and the next call is unaffected"
    with pytest.raises(GoalForceError, match="_pool_scope"):
        vae._padded_buffer(3, 8, 12, 192, z.device)
    side = torch.cuda.Stream()
    with vae._pool_scope():
        vae._padded_buffer(3, 8, 12, 192, z.device)
        with torch.cuda.stream(side), pytest.raises(GoalForceError, match="ONE stream"):
            vae._padded_buffer(3, 8, 12, 192, z.device)
    assert vae._pad_pool == {}


@pytest.mark.gpu
@pytest.mark.parametrize("C", [96, 192, 384])
def test_hip_rmsnorm_silu_three_chunk_kernel_is_bit_identical(C):
    """RMS_norm (+SiLU) at C = 96 / 192 / 384: the all-lanes-live kernel (three chunks per lane, the row sum's two top butterfly stages
    as in-lane additions) against the power-of-two kernel (options(vae_rms3=0)) — bit for bit, ragged row counts included."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(C)
    for rows in (1, 37, 4099, 120 * 208):
        x = (torch.randn((rows, C), generator=g) * 2.5).to(BF).cuda()
        gam = (1 + 0.1 * torch.randn(C, generator=g)).to(BF).cuda()
        for silu in (True, False):
            with ops.options(vae_rms3=0):
                ref = ops.vae_rmsnorm_silu(x, gam, silu=silu)
            got = ops.vae_rmsnorm_silu(x, gam, silu=silu)
            assert torch.equal(got, ref), (C, rows, silu, int((got != ref).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("T,Hs,Ws,t_off", [(3, 12, 16, 0), (5, 8, 48, 1), (2, 120, 208, 0), (9, 20, 32, 2)])
def test_hip_direct_upsample_conv_equals_implicit_gemm(T, Hs, Ws, t_off):
    """The decoder's full-resolution upsample convolution (nearest 2x + 3x3, 192 -> 96 channels): the direct kernel (8 x 32 output
    patches, half-resolution halo, 18 weight stages per frame) against the implicit GEMM (options(conv_direct=0)) and, at the small sizes,
    the patch matrix + GEMM — bit for bit, with a frame offset and over several frame segments."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(T * 1000 + Hs)
    C, N = 192, 96
    k = 9 * C
    x = (torch.randn((T, Hs, Ws, C), generator=g) * 0.7).to(BF).cuda()
    w = (torch.randn((N, k), generator=g) / k ** 0.5).to(BF).cuda()
    b = torch.randn((N,), generator=g).to(BF).cuda()
    kw = dict(upsample2x=True, t_off=t_off, t_out=T - t_off)
    with ops.options(conv_direct=0):
        ref = ops.vae_conv3d(x, None, w, b, 1, 3, **kw)
    got = ops.vae_conv3d(x, None, w, b, 1, 3, **kw)
    assert got.shape == ref.shape == ((T - t_off) * 4 * Hs * Ws, N)
    assert torch.equal(got, ref), f"{int((got != ref).sum())} of {ref.numel()} differ"
    if Hs * Ws < 2000:
        assert torch.equal(ops.gemm(ops.vae_im2col(x, None, 1, 3, k, **kw), w, b), ref)
    assert float(ref.float().abs().max()) > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_hip_implicit_conv_random_shapes(seed):
    """Randomised geometry (frames, image size, channels, output width, kernel, gather mode, temporal stride / offset, history in
    front or separate, residual): every variant of the implicit GEMM == gf_vae_im2col + gf_gemm_bf16 bit for bit."""
    import random
    from goal_force_amd import ops
    rnd = random.Random(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    kt, ks = rnd.choice([(3, 3), (3, 3), (1, 3), (3, 1), (1, 1)])
    mode = rnd.choice([0, 0, 1, 2]) if ks == 3 and kt == 1 else 0
    C = rnd.choice([8, 16, 24, 64, 96, 192])
    N = rnd.choice([8, 24, 96, 128, 136, 192])
    H, W = rnd.randrange(2, 14) * 2, rnd.randrange(2, 14) * 2
    T = rnd.randrange(1, 6)
    t_stride = 2 if (kt == 3 and ks == 1 and T >= 3 and rnd.random() < 0.5) else 1
    t_off = rnd.randrange(0, 2) if T >= 3 else 0
    t_out = max(1, (T - t_off + t_stride - 1) // t_stride - rnd.randrange(0, 2))
    if t_off + (t_out - 1) * t_stride >= T:
        t_out = (T - t_off + t_stride - 1) // t_stride
    gather = dict(upsample2x=(mode == 1), downsample2=(mode == 2), t_stride=t_stride, t_off=t_off, t_out=t_out)
    k = kt * ks * ks * C
    kpad = -(-k // 64) * 64
    buf = torch.randn((T + 2, H, W, C), generator=g).to(BF).cuda()
    x, cache = buf[2:], (buf[:2].clone() if kt == 3 else None)
    w = torch.zeros((N, kpad), dtype=BF)
    w[:, :k] = (torch.randn((N, k), generator=g) / k ** 0.5).to(BF)
    w = w.cuda()
    b = torch.randn((N,), generator=g).to(BF).cuda()
    cols = ops.vae_im2col(x.clone(), cache, kt, ks, kpad, **gather)
    resid = torch.randn((cols.shape[0], N), generator=g).to(BF).cuda() if rnd.random() < 0.5 else None
    ref = ops.gemm(cols, w, b, epilogue=ops.EPI_BIAS if resid is None else ops.EPI_BIAS_RESID, resid=resid)
    got = ops.vae_conv3d(x.clone(), cache, w, b, kt, ks, resid=resid, **gather)
    assert torch.equal(got, ref), (kt, ks, mode, C, N, H, W, T, gather)
    if kt == 3:
        got2 = ops.vae_conv3d(x, None, w, b, kt, ks, resid=resid, history_in_front=True, **gather)
        assert torch.equal(got2, ref), ("history in front", kt, ks, mode, C, N, H, W, T, gather)


@pytest.mark.gpu
def test_hip_implicit_conv_rejects_bad_arguments():
    from goal_force_amd import ops
    from goal_force_amd._lib import GoalForceError
    x = torch.zeros((2, 4, 4, 16), dtype=BF).cuda()
    w = torch.zeros((8, 448), dtype=BF).cuda()
    with pytest.raises(GoalForceError):
        ops.vae_conv3d(x, None, w, None, 3, 3)                 # temporal kernel without its cache
    with pytest.raises(GoalForceError):
        ops.vae_conv3d(x, None, w[:, :100], None, 1, 3)        # K not a multiple of 64 / does not cover the taps
    with pytest.raises(GoalForceError):
        ops.vae_conv3d(x, None, w[:, :192], None, 1, 3, t_off=1, t_out=2)   # frames past the input


@pytest.mark.gpu
@pytest.mark.parametrize("C", [8, 96, 192, 384])
def test_hip_vae_rmsnorm_all_widths(C):
    """Every lanes-per-row variant of the RMS_norm kernel against the eager bf16 ops, including a ragged last wave."""
    from goal_force_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(C)
    xx = (torch.randn((3, 7, 11, C), generator=g) * 3).to(BF)
    gam = (1 + 0.1 * torch.randn(C, generator=g)).to(BF)
    for silu in (True, False):
        ref = F.normalize(xx, dim=-1) * (C ** 0.5) * gam
        ref = F.silu(ref) if silu else ref
        got = ops.vae_rmsnorm_silu(xx.cuda(), gam.cuda(), silu=silu).cpu()
        bad = ((got.view(torch.int16).int() - ref.view(torch.int16).int()).abs() > 1).float().mean()
        assert rel_l2(got.float(), ref.float()) < 3e-3 and float(bad) < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("C", [96, 128, 384])
def test_hip_vae_silu_is_within_one_ulp_of_exact_division(C):
    """the kernels' SiLU is y * rcp(1 + exp2(-y log2 e)) on v_exp_f32 / v_rcp_f32 (about 1 fp32 ulp each), not expf + an
    IEEE division.  Against torch's F.silu on the SAME bf16 inputs (the kernel's own silu = False output; fp32 exp and exact
    division, rounded once) a value near a bf16 rounding boundary may land one bf16 ulp away: counted here on 1.3 M values — never
    more than 1 ulp, and rare (the first number is what DESIGN.md quotes)."""
    from goal_force_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(100 + C)
    x = (torch.randn((4, 30, 13056 // C * 8 // 8, C), generator=g) * 3).to(BF).cuda()
    gam = (1 + 0.1 * torch.randn(C, generator=g)).to(BF).cuda()
    pre = ops.vae_rmsnorm_silu(x, gam, silu=False)
    got = ops.vae_rmsnorm_silu(x, gam, silu=True)
    ref = F.silu(pre)
    d = (got.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    frac = float((d != 0).float().mean())
    print(f"SiLU C={C}: {int((d != 0).sum())} of {d.numel()} values differ from exact-division SiLU ({frac:.2e}), max {int(d.max())} bf16 ulp")
    assert int(d.max()) <= 1 and frac < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("hw,g", [(1560, 1), (392, 3), (8, 2)])
def test_hip_rowmax_neg_is_exact(hw, g):
    from goal_force_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(hw)
    x = (torch.randn((g * hw, hw), generator=gen, device="cuda") * 50).to(torch.bfloat16)
    x[0] = -7.5                                                              # an all-negative row: the maximum is not clipped at 0
    buf = torch.full((g * hw, 448), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.rowmax_neg(x, buf[:, 384])
    assert torch.equal(buf[:, 384], -x.float().amax(dim=1).to(torch.bfloat16))
    assert bool((buf[:, :384] == 3).all()) and bool((buf[:, 385:] == 3).all())      # only that one column is written
    with pytest.raises(ops.GoalForceError):
        ops.rowmax_neg(x, buf[:-1, 384])


@pytest.mark.gpu
def test_hip_vae_frame_attention_keeps_sdpa_precision_at_peaky_scores():
    """The AttentionBlock's softmax (VAE:326-333) on scores whose row offset went through the GEMM's fp32 accumulator: within 2x of
    torch's SDPA (fp32 scores; what the reference calls) against fp64 at every logit scale, where the one-GEMM form (bf16 RAW scores,
    ops.options(vae_attn_offset=False)) is 5-10x off at logit std >= 3.  Production tile: hw = 30*52, C = 384."""
    import torch.nn.functional as F
    from goal_force_amd import ops, vae
    gen = torch.Generator(device="cuda").manual_seed(0)
    hw, C = 1560, 384
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    for qs, bar_one_gemm in ((1.0, None), (3.0, 3.0), (8.0, 5.0)):
        qkv = torch.randn((2, hw, 3 * C), generator=gen, device="cuda")
        qkv[:, :, :C] *= qs
        qkv = qkv.to(torch.bfloat16)
        q, k, v = (qkv[:, :, i * C:(i + 1) * C].double() for i in range(3))
        ref = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), -1) @ v
        sdpa = rel(F.scaled_dot_product_attention(*(qkv[:, None, :, i * C:(i + 1) * C] for i in range(3)))[:, 0], ref)
        got = rel(vae.frame_attention(qkv, C, torch.empty((2, hw, C), dtype=torch.bfloat16, device="cuda")), ref)
        with ops.options(vae_attn_offset=False):
            one = rel(vae.frame_attention(qkv, C, torch.empty((2, hw, C), dtype=torch.bfloat16, device="cuda")), ref)
        print(f"VAE attention, logit std {qs:g}: two-GEMM {got:.2e}  one-GEMM {one:.2e}  torch SDPA {sdpa:.2e}")
        assert got < 2.0 * sdpa + 1e-3, (qs, got, sdpa)
        if bar_one_gemm:
            assert one > bar_one_gemm * got, (qs, one, got)                  # the switch really is the coarser form


@pytest.mark.gpu
def test_hip_vae_decode_vs_reference_golden():
    g, sd, z1, z2 = _fixture()
    vae = _gpu_vae(sd)
    got = vae.decode(z1.cuda(), tiled=False).cpu()
    f32 = torch.from_numpy(g["decode_f32"])
    ref_bf = gi.from_u16(g["decode_bf16"]).float()
    e, e_ref = rel_l2(got.float(), f32), rel_l2(ref_bf, f32)
    assert got.shape == (1, 3, 9, 64, 64)
    assert e < max(1.5e-2, 1.5 * e_ref), f"vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
    # streaming property: the first output frame depends only on the first latent frame
    got1 = vae.decode(z1[:, :, :1].cuda(), tiled=False).cpu()
    assert torch.equal(got1[:, :, 0], got[:, :, 0])


@pytest.mark.gpu
def test_hip_vae_frame_grouping_is_bit_identical_to_streaming():
    """Decoding / encoding the frames after the first in groups == one latent frame (one 4-frame chunk) at a time."""
    g, sd, z1, z2 = _fixture()
    vae = _gpu_vae(sd)
    gen = torch.Generator().manual_seed(3)
    z = torch.randn((1, 16, 6, 8, 8), generator=gen).to(BF).cuda()
    video = (torch.rand((3, 21, 64, 64), generator=gen) * 2 - 1).to(BF).cuda()
    outs, encs = [], []
    for fpc, merge in ((1, False), (2, False), (20, False), (1, True), (3, True), (20, True)):
        # merge_first: the clip's first frame (which skips the temporal resampling) in one chunk with the first group, the shipped
        # default, against a chunk of its own (how the reference streams it)
        vae.frames_per_chunk, vae.merge_first = fpc, merge
        outs.append(vae.decode(z, tiled=False))
        encs.append(vae.encode([video], tiled=False))
    assert outs[0].shape == (1, 3, 21, 64, 64) and encs[0].shape == (1, 16, 6, 8, 8)
    for o, e in zip(outs[1:], encs[1:]):
        assert torch.equal(outs[0], o) and torch.equal(encs[0], e)
    # a single frame (image): the merged chunk degenerates to the first frame alone
    vae.frames_per_chunk, vae.merge_first = 20, True
    one_m = (vae.decode(z[:, :, :1], tiled=False), vae.encode([video[:, :1]], tiled=False))
    vae.merge_first = False
    one_s = (vae.decode(z[:, :, :1], tiled=False), vae.encode([video[:, :1]], tiled=False))
    assert torch.equal(one_m[0], one_s[0]) and torch.equal(one_m[1], one_s[1]) and one_m[0].shape == (1, 3, 1, 64, 64)


@pytest.mark.gpu
def test_hip_vae_tiled_decode_vs_reference_golden():
    g, sd, z1, z2 = _fixture()
    vae = _gpu_vae(sd)
    got = vae.decode(z2.cuda(), tiled=True, tile_size=(8, 8), tile_stride=(4, 4)).cpu()
    f32 = torch.from_numpy(g["tiled_f32"])
    ref_bf = gi.from_u16(g["tiled_bf16"]).float()
    e, e_ref = rel_l2(got.float(), f32), rel_l2(ref_bf, f32)
    assert got.shape == (1, 3, 5, 96, 128) and float(got.float().abs().max()) <= 1.0
    assert e < max(1.5e-2, 1.5 * e_ref), f"vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"


@pytest.mark.gpu
def test_hip_tile_blend_exact_against_oracle_arithmetic():
    """constant tiles through the blend: ramps, bf16 accumulation order and the final divide are exact."""
    from goal_force_amd import ops
    T, H, W, u = 2, 16, 24, 1
    values = torch.zeros((3, T, H, W), dtype=BF, device="cuda")
    weight = torch.zeros((H, W), dtype=BF, device="cuda")
    ref_v = torch.zeros((1, 3, T, H, W), dtype=BF)
    ref_w = torch.zeros((1, 1, T, H, W), dtype=BF)
    g = torch.Generator().manual_seed(3)
    size, stride = (8, 8), (4, 4)
    tasks = []
    for h in range(0, H, stride[0]):
        if h - stride[0] >= 0 and h - stride[0] + size[0] >= H:
            continue
        for w in range(0, W, stride[1]):
            if w - stride[1] >= 0 and w - stride[1] + size[1] >= W:
                continue
            tasks.append((h, h + size[0], w, w + size[1]))
    for (h, h_, w, w_) in tasks:
        tile = torch.randn((T, 8, 8, 8), generator=g).to(BF)
        ops.vae_tile_blend(values, weight, tile.cuda(), h, w, (h == 0, h_ >= H, w == 0, w_ >= W), (4, 4))
        mh = vo._ramp(8, h == 0, h_ >= H, 4)
        mw = vo._ramp(8, w == 0, w_ >= W, 4)
        mask = torch.minimum(mh[:, None].expand(-1, 8), mw[None, :].expand(8, -1)).view(1, 1, 1, 8, 8).to(BF)
        t5 = tile[..., :3].permute(3, 0, 1, 2)[None]
        ref_v[:, :, :, h:h_, w:w_] += t5 * mask
        ref_w[:, :, :, h:h_, w:w_] += mask
    ops.vae_tile_finalize(values, weight)
    ref = (ref_v / ref_w).clamp_(-1, 1)
    assert torch.equal(values.cpu()[None], ref)


@pytest.mark.gpu
def test_hip_vae_encode_vs_reference_golden():
    g, sd, z1, z2 = _fixture()
    v1, v2 = _fixture.videos
    vae = _gpu_vae(sd)
    for key, got in (("encode", vae.encode(v1.cuda(), tiled=False).cpu()),
                     ("encode_tiled", vae.encode(v2.cuda(), tiled=True, tile_size=(8, 8), tile_stride=(4, 4)).cpu())):
        f32 = torch.from_numpy(g[f"{key}_f32"])
        ref_bf = gi.from_u16(g[f"{key}_bf16"]).float()
        e, e_ref = rel_l2(got.float(), f32), rel_l2(ref_bf, f32)
        assert tuple(got.shape) == tuple(f32.shape)
        assert e < max(1.5e-2, 1.5 * e_ref), f"{key}: vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
    # a [F,H,W,3] control video handed over as a permuted view (GF:800) encodes without a copy
    cv = torch.rand((9, 64, 64, 3)).to(BF)
    a = vae.encode([cv.cuda().permute(3, 0, 1, 2)], tiled=False)
    b = vae.encode([cv.permute(3, 0, 1, 2).contiguous().cuda()], tiled=False)
    assert torch.equal(a, b) and tuple(a.shape) == (1, 16, 3, 8, 8)


@pytest.mark.gpu
def test_hip_vae_production_tile_vs_reference_golden():
    """The tile geometry the pipeline really runs (GF:733: 30 x 52 latent units = 240 x 416 pixels; first-frame path + frame
    groups) against g12_vae_tile.npz — the reference's VideoVAE_.decode / .encode on one such tile with 3 latent frames:
    8192 sampled elements (bar: <= max(1.5e-2, 1.5 x the reference-bf16's own distance from fp32)) and every frame's /
    channel's energy (sum of squares within 2 %)."""
    g = np.load(os.path.join(GOLDEN, "g12_vae_tile.npz"))
    g6 = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    sd = gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61)
    z, vid = gi.vae_tile_inputs()
    assert gi.same_checksum(gi.checksum([z, vid]), g["ck_inputs"])
    vae = _gpu_vae(sd)
    dec = vae.decode_tile_channels_last(z[0].cuda())[..., :3].permute(3, 0, 1, 2)[None].float().cpu()   # [1,3,9,240,416], raw
    enc = vae.encode_tile_channels_last(vid[0].cuda()).permute(3, 0, 1, 2)[None].float().cpu()          # [1,16,3,30,52]
    for name, got in (("decode", dec), ("encode", enc)):
        assert tuple(got.shape) == tuple(int(v) for v in g[f"{name}_shape"])
        idx = gi.grad_sample_index(got.numel(), seed=3100 + len(name), k=8192)
        s32 = torch.from_numpy(g[f"{name}_sample_f32"])
        sbf = gi.from_u16(g[f"{name}_sample_bf16"]).float()
        e, e_ref = rel_l2(got.flatten()[idx], s32), rel_l2(sbf, s32)
        assert e < max(1.5e-2, 1.5 * e_ref), f"{name}: sampled rel-L2 {e:.3e} (reference bf16 {e_ref:.3e})"
        ss = got.double().pow(2).sum(dim=(0, 3, 4)).numpy()
        ref = g[f"{name}_sumsq_f32"]
        assert np.all(np.abs(ss - ref) <= 0.02 * ref + 1e-6), f"{name}: per-(channel, frame) energy off by {np.max(np.abs(ss / ref - 1)):.3e}"


@pytest.mark.gpu
def test_hip_tiled_vae_decode_and_encode_all_tiles_within_reference_drift():
    """All 9 tiles of an 832 x 480 video and the blend (GF:733; VAE:1103-1203) — 3 latent frames = 9 pixel frames here (the
    first-frame path and two streamed chunks; most of the test's time is torch-ROCm's fp32 convolutions being set up), the
    whole 21 / 81 in profiles/r06/fullsize_vae_parity.json — on the HIP path against the reference's arithmetic on torch-ROCm's
    kernels (oracle/vae_oracle.py on this GPU, bf16) with the same arithmetic in fp32 as the yardstick.  Bar as for the DiT at
    full size: HIP-vs-fp32 <= 1.25 x (reference bf16 vs fp32); frames: PSNR not more than 0.5 dB below the reference's."""
    import fullsize_vae_parity as fv
    rep = fv.run(grid=(3, 60, 104), log=lambda s: print(s, flush=True))
    assert rep["decode"]["shape"] == [1, 3, 9, 480, 832] and rep["encode"]["shape"] == [1, 16, 3, 60, 104]
    for leg in ("decode", "encode"):
        r = rep[leg]["rel_l2"]
        assert r["hip_bf16_vs_fp32"] <= 1.25 * r["ref_bf16_vs_fp32"], (leg, r)
        assert r["hip_bf16_vs_fp32"] < 3e-2, (leg, r)
    p = rep["decode"]["psnr_db_uint8_frames"]
    assert p["hip_bf16_vs_fp32"] >= p["ref_bf16_vs_fp32"] - 0.5, p
