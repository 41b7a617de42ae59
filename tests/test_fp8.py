"""fp8 Linear (BASELINE config 5 semantics, VRAM:115-151).
CPU: oracle known-answer vectors, and the oracle against tests/golden/g11_fp8_scaled_mm.npz — outputs of
`torch._scaled_mm` run on an MI355X through the reference's call sequence (tests/golden/make_fp8_golden_gpu.py).
GPU: the HIP quantiser bit-exact with that sequence run live (`torch._scaled_mm` itself, same box) and with the oracle;
the scaled-MFMA GEMM within 1 bf16 ulp of `torch._scaled_mm` (accumulation order is the only freedom: every e4m3 x e4m3
product is exact in fp32)."""
import math
import os

import numpy as np

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import fp8_oracle as fo

BF = torch.bfloat16


def test_oracle_known_answers():
    # row 0: max 896 -> scale 2, values /2 are exactly representable in e4m3; row 1: max < 448 -> scale 1
    x = torch.tensor([[896.0, -448.0, 3.0, 0.5] + [0.0] * 124, [1.0, 2.0, -3.0, 4.0] + [0.0] * 124]).to(BF)
    x8, s = fo.quantize_activation(x)
    assert s.flatten().tolist() == [2.0, 1.0]
    assert x8.float()[0, :4].tolist() == [448.0, -224.0, 1.5, 0.25]
    assert x8.float()[1, :4].tolist() == [1.0, 2.0, -3.0, 4.0]
    # e4m3 has 3 mantissa bits: 17 -> 16, 19 -> 20, 18 exact; tiny values flush to 0
    assert torch.tensor([17.0, 19.0, 18.0, 0.0009]).to(fo.FP8).float().tolist() == [16.0, 20.0, 18.0, 0.0]
    w = torch.zeros((8, 128))
    w[0, :4] = torch.tensor([1.0, 1.0, 1.0, 1.0])
    w[1, :4] = torch.tensor([0.5, 0.0, -2.0, 8.0])
    b = torch.arange(8).float()
    out = fo.fp8_linear(x, w.to(BF), b.to(BF))
    # row 0 col 0: (448 - 224 + 1.5 + 0.25) * 2 + 0 = 451.5 -> bf16 452;  col 1: (224 - 3 + 2) * 2 + 1 = 447 -> bf16 448
    assert out[0, 0].item() == 452.0 and out[0, 1].item() == 448.0
    assert out[1, 0].item() == 4.0 and out[1, 1].item() == 0.5 + 6 + 32 + 1


def _ulp_stats(got, ref):
    """fraction of bf16 outputs more than 1 ulp apart, and the rel-L2 distance."""
    d = (got.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    return float((d > 1).float().mean()), rel_l2(got.float(), ref.float())


def test_oracle_vs_scaled_mm_fixture():
    """The oracle's pin: torch._scaled_mm outputs recorded on the GPU (the reference's fp8_linear is that torch call)."""
    import gen_inputs as gi
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "g11_fp8_scaled_mm.npz"))
    for i, (M, N, K) in enumerate(gi.FP8_CASES):
        x, w, b = gi.fp8_case(M, N, K)
        assert gi.same_checksum(gi.checksum([x, w, b]), g[f"ck{i}"])
        x8, s = fo.quantize_activation(x)
        assert np.array_equal(x8.view(torch.uint8).numpy(), g[f"x8_{i}"]), "quantised activations: bit-exact"
        assert np.array_equal(s.numpy(), g[f"s{i}"])
        if g[f"w8_{i}"].size:
            assert np.array_equal(w.to(fo.FP8).view(torch.uint8).numpy(), g[f"w8_{i}"])
        bad, e = _ulp_stats(fo.fp8_linear(x, w, b), gi.from_u16(g[f"y{i}"]))
        assert bad < 1e-3 and e < 1e-3, f"case {i}: >1ulp frac {bad:.2e}, rel-L2 {e:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(72, 256, 256), (300, 528, 384), (1000, 5120, 5120), (257, 13824, 1024)])
def test_hip_fp8_vs_torch_scaled_mm(M, N, K):
    """The reference's fp8_linear run live: torch._scaled_mm on this GPU through the call sequence of VRAM:115-151."""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_fp8_golden_gpu import scaled_mm_linear
    import gen_inputs as gi
    from goal_force_amd import ops
    x, w, b = gi.fp8_case(M, N, K)
    ref, r8, rs, rw8 = scaled_mm_linear(x.cuda(), w.cuda(), b.cuda())
    x8, s = ops.quant_fp8_rowscale(x.cuda())
    assert torch.equal(x8.view(torch.uint8), r8.view(torch.uint8)), "quantised activations must be bit-exact"
    assert torch.equal(s, rs.flatten())
    w8 = ops.cast_fp8(w.cuda())
    assert torch.equal(w8.view(torch.uint8), rw8.view(torch.uint8))
    got = ops.gemm_fp8(x8, s, w8, b.cuda())
    bad, e = _ulp_stats(got.cpu(), ref.cpu())
    assert bad < 1e-3 and e < 1e-3, f"vs torch._scaled_mm: >1ulp frac {bad:.2e}, rel-L2 {e:.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(72, 256, 256), (300, 520, 384), (1000, 5120, 5120), (257, 13824, 1024)])
def test_hip_fp8_linear_vs_oracle(M, N, K):
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn((M, K), generator=g) * 3).to(BF)
    x[M // 2, 5] = 1500.0   # forces scale_a > 1 on one row
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(BF)
    b = (0.1 * torch.randn(N, generator=g)).to(BF)
    x8, s = ops.quant_fp8_rowscale(x.cuda())
    r8, rs = fo.quantize_activation(x)
    assert torch.equal(x8.cpu().view(torch.uint8), r8.view(torch.uint8)), "quantised activations must be bit-exact"
    assert torch.equal(s.cpu(), rs.flatten())
    w8 = ops.cast_fp8(w.cuda())
    assert torch.equal(w8.cpu().view(torch.uint8), w.to(fo.FP8).view(torch.uint8))
    got = ops.gemm_fp8(x8, s, w8, b.cuda()).cpu()
    ref = fo.fp8_linear(x, w, b)
    e = rel_l2(got.float(), ref.float())
    bad = ((got.view(torch.int16).int() - ref.view(torch.int16).int()).abs() > 1).float().mean()
    assert e < 1e-3 and float(bad) < 1e-3, f"rel_l2={e:.3e}, >1ulp frac={float(bad):.2e}"
    # fp8 contract vs exact bf16 linear: the quantisation error itself (documented, not a kernel property)
    e_q = rel_l2(got.float(), F.linear(x.float(), w.float(), b.float()))
    assert e_q < 8e-2


@pytest.mark.gpu
def test_hip_fp8_epilogues_match_bf16_kernel_semantics():
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K = 300, 256, 256
    x = torch.randn((M, K), generator=g).to(BF)
    w = (torch.randn((N, K), generator=g) / 16).to(BF)
    b, gate = torch.randn(N, generator=g).to(BF), torch.randn(N, generator=g).to(BF)
    r = torch.randn((M, N), generator=g).to(BF)
    y = fo.fp8_linear(x, w, b)
    x8, s = ops.quant_fp8_rowscale(x.cuda())
    w8 = ops.cast_fp8(w.cuda())
    for epi, ref in ((ops.EPI_BIAS_GELU_TANH, F.gelu(y, approximate="tanh")), (ops.EPI_BIAS_RESID, r + y),
                     (ops.EPI_BIAS_GATE_RESID, r + gate * y)):
        got = ops.gemm_fp8(x8, s, w8, b.cuda(), epilogue=epi, resid=r.cuda(), gate=gate.cuda()).cpu()
        assert rel_l2(got.float(), ref.float()) < 2e-3


@pytest.mark.gpu
def test_hip_fp8_dit_block_vs_fp8_oracle_block():
    """Config 5 at block level: every Linear of the block on the fp8_linear contract (HIP) vs the oracle block with
    the restated fp8_linear swapped in for F.linear (same bf16 graph otherwise)."""
    import gen_inputs as gi
    from oracle import wan_oracle as wo
    from goal_force_amd.dit import DiTBlock, RopeTable, enable_fp8, precompute_freqs_cis_3d
    cfg = gi.MID
    grid = (2, 6, 8)
    S = 96
    sd = gi.block_sd(torch.Generator().manual_seed(21), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], S, 64, seed=22)
    blk = DiTBlock(False, cfg["dim"], cfg["num_heads"], cfg["ffn_dim"], cfg["eps"])
    blk.load_state_dict(sd, strict=True)
    blk = enable_fp8(blk.to(BF).cuda())
    rope = RopeTable.from_grid(precompute_freqs_cis_3d(128), *grid, "cuda")
    got = blk(x.cuda(), ctx.cuda(), t_mod.cuda(), rope).cpu()
    freqs = wo.rope_freqs_3d(128, *grid)
    exact = wo.dit_block(x.float(), ctx.float(), t_mod.float(), freqs, {k: v.float() for k, v in sd.items()}, "",
                         cfg["num_heads"], cfg["eps"])
    old = wo.LINEAR
    wo.LINEAR = fo.fp8_linear
    try:
        ref = wo.dit_block(x, ctx, t_mod, freqs, sd, "", cfg["num_heads"], cfg["eps"])
    finally:
        wo.LINEAR = old
    e = rel_l2(got.float(), ref.float())
    e_q = rel_l2(ref.float(), exact)
    # bf16-level differences between the two paths flip individual e4m3 roundings, so they differ by a fraction
    # of the fp8 contract's own quantisation noise e_q; bar: < 0.5 * e_q
    assert e < 0.5 * e_q, f"HIP fp8 block vs fp8 oracle block {e:.3e} (fp8 contract itself is {e_q:.3e} from fp32 math)"
    # turning fp8 off restores the bf16 path
    enable_fp8(blk, False)
    got_bf = blk(x.cuda(), ctx.cuda(), t_mod.cuda(), rope).cpu()
    assert rel_l2(got_bf.float(), exact) < 5e-3


# ------------------------------------------------------------------ the 4-wave fp8 kernel (gemm_a4_kernel<EPI, true>, M >= 512)
@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(700, 520, 384), (512, 256, 128), (1000, 1024, 256), (2300, 264, 1280)])
def test_hip_fp8_a4_exact_integer_layout(M, N, K):
    """Small-integer e4m3 operands (every product and fp32 sum exact, so the MFMA's internal order cannot matter): any fragment /
    chunk-pairing / half-tile / register-set mistake of the asm loop is a hard mismatch.  K = 128 (one tile: the body's second tile
    multiplies a zero tile), 384 (odd tile count), ragged M and N tiles; rows scaled by exact powers of two."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    w[:, 0] += (torch.arange(N) % 3).float()            # asymmetric
    a[:, K - 1] += (torch.arange(M) % 2).float()
    bias = torch.randint(-4, 5, (N,), generator=g).float()
    rs = (2.0 ** torch.randint(0, 3, (M,), generator=g)).float()
    ref = (a @ w.t()) * rs[:, None] + bias
    assert ref.abs().max() < 2048
    ref = ref.to(BF)                                     # |ref| may exceed 256: the kernel's only rounding is the final one
    a8, w8 = a.cuda().to(torch.float8_e4m3fn), w.cuda().to(torch.float8_e4m3fn)
    assert torch.equal(a8.float().cpu(), a) and torch.equal(w8.float().cpu(), w)
    got = ops.gemm_fp8(a8, rs.cuda(), w8, bias.to(BF).cuda())
    assert torch.equal(got.cpu(), ref), f"{int((got.cpu() != ref).sum())} of {ref.numel()} elements differ"
    with ops.options(prefer_8wave=1):           # the 8-wave kernel on the same operands: same exact result
        assert torch.equal(ops.gemm_fp8(a8, rs.cuda(), w8, bias.to(BF).cuda()).cpu(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1000, 5120, 5120), (700, 520, 384), (2300, 264, 13824), (513, 13824, 256)])
def test_hip_fp8_a4_vs_8wave_kernel(M, N, K):
    """Random data, every fused epilogue: the 4-wave kernel against the 8-wave one-barrier kernel.  Both add the same exact e4m3
    products in fp32; the order differs (k chunks paired (c, c + 4) instead of (2c, 2c + 1), rotated K start), so: <= 1 bf16 ulp
    on all but a sliver.  A row's bits do not depend on the M tiling; strided A and C."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(M + 3 * N + K)
    x = (torch.randn((M, K), generator=g) * 2).to(BF).cuda()
    w8 = ops.cast_fp8((torch.randn((N, K), generator=g) / math.sqrt(K)).to(BF).cuda())
    bias = (0.1 * torch.randn(N, generator=g)).to(BF).cuda()
    resid = torch.randn((M, N), generator=g).to(BF).cuda()
    gate = torch.randn(N, generator=g).to(BF).cuda()
    x8, s = ops.quant_fp8_rowscale(x)
    cases = [dict(), dict(epilogue=ops.EPI_BIAS_GELU_TANH), dict(epilogue=ops.EPI_BIAS_SILU),
             dict(epilogue=ops.EPI_BIAS_RESID, resid=resid), dict(epilogue=ops.EPI_BIAS_GATE_RESID, resid=resid, gate=gate),
             dict(epilogue=ops.EPI_BIAS_MUL, resid=resid)]
    for kw in cases:
        with ops.options(prefer_8wave=1):
            want = ops.gemm_fp8(x8, s, w8, bias, **kw)
        got = ops.gemm_fp8(x8, s, w8, bias, **kw)
        bad, e = _ulp_stats(got.cpu(), want.cpu())
        assert bad < 2e-3 and e < 1e-3, f"{kw.get('epilogue')}: >1ulp frac {bad:.2e}, rel-L2 {e:.3e}"
    full = ops.gemm_fp8(x8, s, w8, bias)
    assert torch.equal(ops.gemm_fp8(x8, s, w8, None), ops.gemm_fp8(x8, s, w8, torch.zeros_like(bias)))
    assert torch.equal(ops.gemm_fp8(x8[M - 512:], s[M - 512:].contiguous(), w8, bias), full[M - 512:])
    wide = torch.zeros((M, K + 128), dtype=torch.float8_e4m3fn, device="cuda")
    wide[:, 128:] = x8
    big = torch.zeros((M, N + 16), dtype=BF, device="cuda")
    ops.gemm_fp8(wide[:, 128:], s, w8, bias, out=big[:, 8:8 + N])
    assert torch.equal(big[:, 8:8 + N], full)
    assert float(big[:, :8].abs().sum()) == 0 and float(big[:, 8 + N:].abs().sum()) == 0


@pytest.mark.gpu
def test_hip_fp8_a4_full_size_vs_torch_scaled_mm():
    """BASELINE config 5's GEMM shapes at S = 32760 (ragged last M tile): the 4-wave kernel against torch._scaled_mm run live
    through the reference's call sequence — quantiser bit-exact, outputs within 1 bf16 ulp, every row finite."""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_fp8_golden_gpu import scaled_mm_linear
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(11)
    S = 32760
    for (N, K) in ((5120, 5120), (13824, 5120), (5120, 13824)):
        x = (torch.randn((S, K), generator=g) * 1.5).to(BF).cuda()
        x[77, 5] = 2000.0                                 # one row with scale_a > 1
        w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(BF).cuda()
        b = (0.1 * torch.randn(N, generator=g)).to(BF).cuda()
        ref, r8, rs, rw8 = scaled_mm_linear(x, w, b)
        x8, s = ops.quant_fp8_rowscale(x)
        assert torch.equal(x8.view(torch.uint8), r8.view(torch.uint8)) and torch.equal(s, rs.flatten())
        got = ops.gemm_fp8(x8, s, ops.cast_fp8(w), b)
        assert bool(torch.isfinite(got.float().norm(dim=1)).all())
        bad, e = _ulp_stats(got.cpu(), ref.cpu())
        assert bad < 1e-3 and e < 1e-3, f"[{S},{K}]x[{N},{K}]^T vs torch._scaled_mm: >1ulp frac {bad:.2e}, rel-L2 {e:.3e}"
        del x, w, ref, r8, x8, got


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [5120, 4096, 1536])
@pytest.mark.parametrize("kind", ["modulate", "affine", "plain"])
def test_hip_layernorm_fp8_fusion_is_bit_identical(dim, kind):
    """gf_layernorm_modulate_fp8 (the normalised row quantised in the registers of the wave that normalised it) against the two
    kernels it fuses — e4m3 bytes and scales bit-identical, including a row whose maximum exceeds 448 (scale_a > 1) and a
    constant row (variance 0)."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(dim)
    M = 777
    x = (torch.randn((M, dim), generator=g) * 2).to(BF)
    x[5, 17] = 3000.0
    x[9] = 1.25
    kw = {}
    if kind == "modulate":
        kw = dict(scale1p=(1 + 0.3 * torch.randn(dim, generator=g)).to(BF).cuda(), shift=(0.2 * torch.randn(dim, generator=g)).to(BF).cuda())
        kw["scale1p"][3] = 400.0         # drives one output column of every row far up: row maxima above 448 (scale_a > 1)
    elif kind == "affine":
        kw = dict(weight=(1 + 0.2 * torch.randn(dim, generator=g)).to(BF).cuda(), bias=(0.1 * torch.randn(dim, generator=g)).to(BF).cuda())
    xd = x.cuda()
    want8, wants = ops.quant_fp8_rowscale(ops.layernorm_modulate(xd, **kw))
    got8, gots = ops.layernorm_modulate_fp8(xd, **kw)
    assert torch.equal(gots, wants) and float(wants.max()) >= 1.0
    assert torch.equal(got8.view(torch.uint8), want8.view(torch.uint8))
    if kind == "modulate":
        assert float(wants.max()) > 1.0, "the case must exercise scale_a > 1"


@pytest.mark.gpu
@pytest.mark.parametrize("skv,n,k", [(2100, 512, 256), (4100, 1024, 384), (32760, 5120, 5120)])
def test_hip_fp8_linear_vt32_is_projection_plus_transpose_bit_for_bit(skv, n, k):
    """gf_linear_vt32_fp8 (config 5's V projection written as attention kernel 3's V^T operand: the 4-wave fp8 GEMM with the
    operands swapped, the tokens' activation scales applied per COLUMN) against the path it replaces, gf_gemm_fp8 then
    gf_transpose_v32 — the same bits, zero key columns from kv_len to kv_pad included; and SelfAttention.attend under enable_fp8
    with it == with the plain fp8 projection + transpose (GF_VT_FROM_GEMM=0)."""
    from goal_force_amd import _lib, ops
    g = torch.Generator().manual_seed(skv + n)
    x = (torch.randn((skv, k), generator=g) * 2).to(BF).cuda()
    x[11, 3] = 1800.0                                                      # one token with scale_a > 1
    w8 = ops.cast_fp8((torch.randn((n, k), generator=g) * 0.05).to(BF).cuda())
    b = torch.randn((n,), generator=g).to(BF).cuda()
    x8, s = ops.quant_fp8_rowscale(x)
    kv_pad = -(-skv // 64) * 64
    v = ops.gemm_fp8(x8, s, w8, b)
    want = torch.full((n * kv_pad,), 7.0, dtype=BF, device="cuda")
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.gf_transpose_v32(v.data_ptr(), v.stride(0), want.data_ptr(), skv, kv_pad, n // 128, st) == 0
    got = ops.linear_vt32_fp8(x8, s, w8, b)[: n * kv_pad].clone()
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    if n == 512:
        from goal_force_amd.dit import RopeTable, SelfAttention, enable_fp8, DiTBlock
        torch.manual_seed(3)
        blk = enable_fp8(DiTBlock(False, 512, 4, 1024).to(BF).cuda())
        sa = blk.self_attn
        xa = (torch.randn((skv, 512), generator=g) * 0.5).to(BF).cuda()
        pos = torch.arange(skv, dtype=torch.float32)[:, None] * torch.arange(1, 65, dtype=torch.float32)[None, :] * 1e-3
        rope = RopeTable(torch.polar(torch.ones_like(pos), pos), "cuda")
        a = sa.attend(xa, rope)
        with ops.options(vt_from_gemm=False):
            bb = sa.attend(xa, rope)
        assert torch.equal(a, bb)
