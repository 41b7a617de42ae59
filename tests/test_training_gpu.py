"""GPU parity of the training-step kernels against torch autograd in fp32 on the same bf16 inputs — the
arithmetic the reference's loss.backward() runs (training_loss, src/goal_force/wan_video_new.py:180-193; attention
DIT:28-61).  Tolerances: bf16 gradients of bf16 graphs, fp32 accumulation: rel-L2 <= 1e-2 against the fp32 result."""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
HD = 128


@pytest.fixture(autouse=True)
def _grad_enabled():
    """Other test modules switch autograd off process-wide at import; the training tests need the tape."""
    with torch.enable_grad():
        yield


def _ref_attention_grads(q, k, v, dout, heads):
    qf, kf, vf = (t.float().cpu().requires_grad_(True) for t in (q, k, v))
    sq, skv = q.shape[0], k.shape[0]
    qh = qf.view(sq, heads, HD).transpose(0, 1)
    kh = kf.view(skv, heads, HD).transpose(0, 1)
    vh = vf.view(skv, heads, HD).transpose(0, 1)
    s = qh @ kh.transpose(1, 2) / math.sqrt(HD)
    o = (torch.softmax(s, dim=-1) @ vh).transpose(0, 1).reshape(sq, heads * HD)
    o.backward(dout.float().cpu())
    lse2 = torch.logsumexp(s, dim=-1).transpose(0, 1) * math.log2(math.e)      # log2 domain, [sq, heads]
    return o.detach(), lse2.detach(), qf.grad, kf.grad, vf.grad


@pytest.mark.parametrize("sq,skv,heads", [(72, 72, 2), (300, 200, 3), (1000, 512, 4), (128, 64, 1), (257, 130, 2), (5, 3, 1), (33, 129, 8),
                                          (31, 2000, 2)])
def test_flash_attn_backward(sq, skv, heads):
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(sq * 7 + skv)
    q = torch.randn((sq, heads * HD), generator=g).to(BF).cuda()
    k = torch.randn((skv, heads * HD), generator=g).to(BF).cuda()
    v = torch.randn((skv, heads * HD), generator=g).to(BF).cuda()
    dout = torch.randn((sq, heads * HD), generator=g).to(BF).cuda()
    o, lse = ops.flash_attn_lse(q, k, v, heads)
    assert torch.equal(o, ops.flash_attn(q, k, v, heads)), "the lse variant must not change the forward result"
    o_ref, lse_ref, dq_ref, dk_ref, dv_ref = _ref_attention_grads(q, k, v, dout, heads)
    assert rel_l2(o.float().cpu(), o_ref) < 4e-3
    assert float((lse.cpu() - lse_ref).abs().max()) < 2e-3
    dq, dk, dv = ops.flash_attn_bwd(q, k, v, o, dout, lse, heads)
    for name, got, want in (("dq", dq, dq_ref), ("dk", dk, dk_ref), ("dv", dv, dv_ref)):
        e = rel_l2(got.float().cpu(), want)
        assert e < 1e-2, f"{name}: rel_l2={e:.3e}"
        assert got.dtype == BF and got.shape == want.shape


@pytest.mark.parametrize("sq,skv,heads", [(1000, 777, 8), (2085, 1999, 8), (96, 4000, 3)])
def test_flash_attn_backward_ragged_lengths_and_xcd_grid_vs_fp32_autograd(sq, skv, heads):
    """dQ (32 queries per wave) and the paired dK / dV kernel (48 keys per wave pair) against fp32 autograd of the written-out
    attention at ragged lengths and on the XCD-ordered grid (heads % 8 == 0).  Bar: 5e-3 rel-L2 per gradient — the bf16 rounding of
    the outputs alone is ~2e-3; the shipped kernels measured 2.4e-3 on dQ (profiles/r04/attnbwd_ab_qscale.log), and the variant
    that rebuilt P from the forward's pre-scaled Q' (3.0e-3, tools/patches/attention_bwd_experiments.patch) was dropped for it."""
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(sq + skv)
    q = torch.randn((sq, heads * HD), generator=g).to(BF).cuda()
    k = torch.randn((skv, heads * HD), generator=g).to(BF).cuda()
    v = torch.randn((skv, heads * HD), generator=g).to(BF).cuda()
    dout = torch.randn((sq, heads * HD), generator=g).to(BF).cuda()
    o, lse = ops.flash_attn_lse(q, k, v, heads)
    got = ops.flash_attn_bwd(q, k, v, o, dout, lse, heads)
    ref = _ref_attention_grads(q, k, v, dout, heads)[2:]
    for name, a, r in zip(("dq", "dk", "dv"), got, ref):
        assert torch.isfinite(a.float()).all(), name
        e = rel_l2(a.float().cpu(), r)
        assert e < 5e-3, f"{name}: vs fp32 autograd {e:.3e}"


def test_flash_attn_backward_strided_inputs():
    """q, k, v as column slices of one fused [S, 3D] buffer (row stride 3D), as the fused QKV projection produces them."""
    from goal_force_amd import ops
    heads, s = 2, 200
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn((s, 3 * heads * HD), generator=g).to(BF).cuda()
    q, k, v = qkv[:, :heads * HD], qkv[:, heads * HD:2 * heads * HD], qkv[:, 2 * heads * HD:]
    dout = torch.randn((s, heads * HD), generator=g).to(BF).cuda()
    o, lse = ops.flash_attn_lse(q, k, v, heads)
    dq, dk, dv = ops.flash_attn_bwd(q, k, v, o, dout, lse, heads)
    dq2, dk2, dv2 = ops.flash_attn_bwd(q.contiguous(), k.contiguous(), v.contiguous(), o, dout, lse, heads)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    dq3, dk3, dv3 = ops.flash_attn_bwd(q, k, v, o, dout, lse, heads, need_dkv=False)      # dq alone: the dK/dV kernel is not launched
    assert torch.equal(dq3, dq) and dk3 is None and dv3 is None


# ---------------------------------------------------------------------------------------------------------------------
# row / elementwise backward kernels, loss, optimiser
def _bf(t):
    return t.to(BF).cuda()


def test_layernorm_backward_modulate_and_affine():
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(1)
    rows, dim = 75, 256
    x, dy = torch.randn((rows, dim), generator=g).to(BF), torch.randn((rows, dim), generator=g).to(BF)
    gmul = (1 + 0.1 * torch.randn((dim,), generator=g)).to(BF)
    shift = (0.1 * torch.randn((dim,), generator=g)).to(BF)
    xf, gf, bf_ = x.float().requires_grad_(True), gmul.float().requires_grad_(True), shift.float().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xf, (dim,), None, None, 1e-6) * gf + bf_
    y.backward(dy.float())
    dg, db = torch.zeros(dim, device="cuda"), torch.zeros(dim, device="cuda")
    dx = ops.layernorm_bwd(_bf(x), _bf(dy), g=_bf(gmul), dg_acc=dg, db_acc=db, eps=1e-6)
    assert rel_l2(dx.float().cpu(), xf.grad) < 6e-3
    assert rel_l2(dg.cpu(), gf.grad) < 2e-3 and rel_l2(db.cpu(), bf_.grad) < 1e-5
    # no multiplier, no accumulators
    xf2 = x.float().requires_grad_(True)
    torch.nn.functional.layer_norm(xf2, (dim,), None, None, 1e-6).backward(dy.float())
    assert rel_l2(ops.layernorm_bwd(_bf(x), _bf(dy)).float().cpu(), xf2.grad) < 6e-3


@pytest.mark.parametrize("rope", [True, False])
def test_rmsnorm_rope_backward(rope):
    from goal_force_amd import ops
    from oracle import wan_oracle as wo
    g = torch.Generator().manual_seed(2)
    rows, heads, hd = 72, 2, 128
    dim = heads * hd
    x, dy = torch.randn((rows, dim), generator=g).to(BF), torch.randn((rows, dim), generator=g).to(BF)
    w = (1 + 0.1 * torch.randn((dim,), generator=g)).to(BF)
    freqs = wo.rope_freqs_3d(hd, 3, 4, 6)                                  # [72, 64] complex
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * wf
    if rope:
        yc = torch.view_as_complex(y.double().reshape(rows, heads, hd // 2, 2)) * freqs[:, None, :]
        y = torch.view_as_real(yc).flatten(1).float()
    y.backward(dy.float())
    cos = freqs.real.float().contiguous().cuda() if rope else None
    sin = freqs.imag.float().contiguous().cuda() if rope else None
    dw = torch.zeros(dim, device="cuda")
    dx = ops.rmsnorm_rope_bwd(_bf(x), _bf(dy), _bf(w), cos, sin, hd, 1e-6, dw_acc=dw)
    assert rel_l2(dx.float().cpu(), xf.grad) < 6e-3
    assert rel_l2(dw.cpu(), wf.grad) < 2e-3


def test_colsum_gate_act_bwd_mse_adamw():
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(4)
    rows, dim = 130, 512
    a, b = torch.randn((rows, dim), generator=g).to(BF), torch.randn((rows, dim), generator=g).to(BF)
    gate = torch.randn((dim,), generator=g).to(BF)
    acc = torch.zeros(dim, device="cuda")
    out = ops.colsum(_bf(a), b=_bf(b), gate=_bf(gate), acc=acc)
    assert rel_l2(acc.cpu(), (a.float() * b.float()).sum(0)) < 1e-5
    assert torch.equal(out.cpu(), (a.float() * gate.float()).to(BF))
    acc2 = torch.zeros(dim, device="cuda")
    assert ops.colsum(_bf(a), acc=acc2) is None and rel_l2(acc2.cpu(), a.float().sum(0)) < 1e-5
    # activations
    for kind, fn in (("gelu_tanh", lambda t: torch.nn.functional.gelu(t, approximate="tanh")), ("silu", torch.nn.functional.silu)):
        uf = a.float().requires_grad_(True)
        fn(uf).backward(b.float())
        assert rel_l2(ops.act_bwd(_bf(a), _bf(b), kind).float().cpu(), uf.grad) < 4e-3
    # loss
    pf = a.float().requires_grad_(True)
    l_ref = torch.nn.functional.mse_loss(pf, b.float()) * 0.37
    l_ref.backward()
    loss, dpred = ops.mse_loss(_bf(a), _bf(b), weight=0.37)
    assert abs(float(loss) - float(l_ref.detach())) < 1e-5 * float(l_ref.detach())
    assert rel_l2(dpred.float().cpu(), pf.grad) < 4e-3
    # AdamW against torch.optim.AdamW on fp32 copies (3 steps)
    p0, grads = torch.randn((1000,), generator=g).to(BF), [torch.randn((1000,), generator=g).to(BF) for _ in range(3)]
    pr = torch.nn.Parameter(p0.float().clone())
    opt = torch.optim.AdamW([pr], lr=1e-2, weight_decay=1e-2)
    p, m, v = _bf(p0), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for i, gr in enumerate(grads):
        pr.grad = gr.float()
        opt.step()
        ops.adamw_step(p, _bf(gr), m, v, i + 1, lr=1e-2, weight_decay=1e-2)
    # bf16 parameter storage: up to half an ulp (2^-9 relative) of rounding per step on top of the fp32 trajectory
    assert bool(((p.float().cpu() - pr.detach()).abs() <= 3 * 2.0 ** -8 * pr.detach().abs() + 1e-3).all())
    assert rel_l2(m.cpu(), opt.state[pr]["exp_avg"]) < 1e-5 and rel_l2(v.cpu(), opt.state[pr]["exp_avg_sq"]) < 1e-4


def test_linear_backward():
    from goal_force_amd.training import LinearFn
    g = torch.Generator().manual_seed(5)
    m, k, n = 72, 256, 512
    x, w, b = torch.randn((m, k), generator=g).to(BF), (torch.randn((n, k), generator=g) / 16).to(BF), torch.randn((n,), generator=g).to(BF)
    dy = torch.randn((m, n), generator=g).to(BF)
    xf, wf, bf_ = (t.float().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.linear(xf, wf, bf_).backward(dy.float())
    xc, wc, bc = (_bf(t).requires_grad_(True) for t in (x, w, b))
    LinearFn.apply(xc, wc, bc).backward(_bf(dy))
    for got, want in ((xc.grad, xf.grad), (wc.grad, wf.grad), (bc.grad, bf_.grad)):
        assert rel_l2(got.float().cpu(), want) < 4e-3


# ---------------------------------------------------------------------------------------------------------------------
# block and full training step against the oracle / the reference golden
def _oracle_block_grads(cfg, sd, x, ctx, t_mod, dout):
    from oracle import wan_oracle as wo
    sdf = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    xf = x.float().requires_grad_(True)
    freqs = wo.rope_freqs_3d(cfg["dim"] // cfg["num_heads"], 3, 4, 6)
    y = wo.dit_block(xf, ctx.float(), t_mod.float(), freqs, sdf, "", cfg["num_heads"], cfg["eps"])
    y.backward(dout.float())
    return xf.grad, {k: v.grad for k, v in sdf.items()}


def test_dit_block_backward_vs_oracle():
    import gen_inputs as gi
    from goal_force_amd.dit import DiTBlock, RopeTable, precompute_freqs_cis_3d
    from goal_force_amd.training import block_forward
    cfg = gi.TINY
    sd = gi.block_sd(torch.Generator().manual_seed(9), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 72, gi.TINY_CTX_LEN, seed=12)
    dout = torch.randn(x.shape, generator=torch.Generator().manual_seed(13)).to(BF)
    dx_ref, g_ref = _oracle_block_grads(cfg, sd, x, ctx, t_mod, dout)
    blk = DiTBlock(False, cfg["dim"], cfg["num_heads"], cfg["ffn_dim"], cfg["eps"])
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(BF).cuda()
    rope = RopeTable.from_grid(precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"]), 3, 4, 6, "cuda")
    xc = x[0].cuda().requires_grad_(True)
    with torch.enable_grad():
        y = block_forward(blk, xc, ctx[0].cuda(), t_mod.cuda(), rope)
        y.backward(dout[0].cuda())
    assert rel_l2(xc.grad.float().cpu(), dx_ref[0]) < 1.5e-2, "dx"
    named = dict(blk.named_parameters())
    worst = 0.0
    for n, want in g_ref.items():
        got = named[n].grad
        assert got is not None, n
        e = rel_l2(got.float().cpu().reshape(want.shape), want)
        worst = max(worst, e)
        # a bias on the keys shifts every score of a query by the same amount, to which softmax is blind: what is left of
        # that gradient comes through the RMSNorm only and is cancellation-dominated (its own bf16 noise is ~5e-2)
        bar = 8e-2 if n.endswith("attn.k.bias") else 2.5e-2
        assert e < bar, f"{n}: rel_l2={e:.3e}"
    # a frozen block back-propagates activations only and gives the same dx
    for p_ in blk.parameters():
        p_.requires_grad_(False)
        p_.grad = None
    xc2 = x[0].cuda().requires_grad_(True)
    with torch.enable_grad():
        block_forward(blk, xc2, ctx[0].cuda(), t_mod.cuda(), rope).backward(dout[0].cuda())
    # (not bit for bit: without column sums the row-norm backward kernels are the wave-per-row forms, which add a row in another order)
    assert rel_l2(xc2.grad.float().cpu(), xc.grad.float().cpu()) < 2e-3 and all(p_.grad is None for p_ in blk.parameters())


def test_dit_block_backward_mid_size_vs_oracle():
    """Wan-1.3B block shape (D=1536, 12 heads of 128, FFN 8960, 960 tokens = grid 5x12x16, 512 text tokens): widths and row
    counts that exercise the multi-chunk row kernels, the M-padding of the weight-gradient GEMMs and 12-head attention."""
    import gen_inputs as gi
    from goal_force_amd.dit import DiTBlock, RopeTable, precompute_freqs_cis_3d
    from goal_force_amd.training import block_forward
    from oracle import wan_oracle as wo
    cfg = gi.MID
    sd = gi.block_sd(torch.Generator().manual_seed(21), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 960, 512, seed=22)
    dout = torch.randn(x.shape, generator=torch.Generator().manual_seed(23)).to(BF)
    sdf = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    xf = x.float().requires_grad_(True)
    freqs = wo.rope_freqs_3d(cfg["dim"] // cfg["num_heads"], 5, 12, 16)
    wo.dit_block(xf, ctx.float(), t_mod.float(), freqs, sdf, "", cfg["num_heads"], cfg["eps"]).backward(dout.float())
    blk = DiTBlock(False, cfg["dim"], cfg["num_heads"], cfg["ffn_dim"], cfg["eps"])
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(BF).cuda()
    rope = RopeTable.from_grid(precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"]), 5, 12, 16, "cuda")
    xc = x[0].cuda().requires_grad_(True)
    block_forward(blk, xc, ctx[0].cuda(), t_mod.cuda(), rope).backward(dout[0].cuda())
    assert rel_l2(xc.grad.float().cpu(), xf.grad[0]) < 1.5e-2, "dx"
    named = dict(blk.named_parameters())
    for n, v in sdf.items():
        e = rel_l2(named[n].grad.float().cpu().reshape(v.grad.shape), v.grad)
        assert e < (8e-2 if n.endswith("attn.k.bias") else 2.5e-2), f"{n}: rel_l2={e:.3e}"


def test_attention_backward_on_the_prescaled_q_keeps_sdpa_precision_at_peaky_logits():
    """Forward (lse) + backward on q' = bf16(c q) with scale = ln 2 — what SelfAttention.attend / DiTBlockFn hand the kernels — against
    fp64 autograd beside torch's bf16 SDPA under autograd, at logit std 8: dQ / dK / dV within 1.25 x SDPA's distance.  On a plain q
    (rounds 1-5: the forward rounds Q' a second time and the backward rebuilds P from other scores than the lse's) the same kernels
    are 3-6 x further away: the test holds that signature too, so that it cannot pass on the wrong path."""
    import torch.nn.functional as F
    from goal_force_amd import ops
    from goal_force_amd.dit import Q_PRESCALE
    S, H, c = ops.VT_MIN_KV, 4, Q_PRESCALE(HD)          # the key length from which the forward runs on kernel 3 (the one that pre-scales Q)
    g = torch.Generator(device="cuda").manual_seed(5)
    q32 = torch.randn((S, H * HD), generator=g, device="cuda") * 8.0
    k, v, do = (torch.randn((S, H * HD), generator=g, device="cuda").to(BF) for _ in range(3))
    heads = lambda t: t.view(S, H, HD).transpose(0, 1)

    def fp64(q):
        q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
        with torch.enable_grad():
            p = torch.softmax(heads(q64) @ heads(k64).transpose(1, 2) / math.sqrt(HD), -1)
            (p @ heads(v64)).transpose(0, 1).reshape(S, H * HD).backward(do.double())
        return q64.grad, k64.grad, v64.grad
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    q = q32.to(BF)
    ref = fp64(q)
    qb, kb, vb = (t.clone().requires_grad_(True) for t in (q, k, v))
    with torch.enable_grad():
        F.scaled_dot_product_attention(heads(qb)[None], heads(kb)[None], heads(vb)[None])[0].transpose(0, 1).reshape(S, H * HD).backward(do)
    sdpa = [rel(a.grad, r) for a, r in zip((qb, kb, vb), ref)]
    o, lse = ops.flash_attn_lse(q, k, v, H)
    plain = [rel(a, r) for a, r in zip(ops.flash_attn_bwd(q, k, v, o, do, lse, H), ref)]
    qp = (q32 * c).to(BF)
    refp = fp64(qp.double() / c)
    o, lse = ops.flash_attn_lse(qp, k, v, H, scale=math.log(2.0))
    dqp, dk, dv = ops.flash_attn_bwd(qp, k, v, o, do, lse, H, scale=math.log(2.0))
    pre = [rel(dqp.double() * c, refp[0]), rel(dk, refp[1]), rel(dv, refp[2])]
    print(f"attention backward at logit std 8 (dq, dk, dv vs fp64): pre-scaled q {pre}  plain q {plain}  torch SDPA {sdpa}")
    for a, b in zip(pre, sdpa):
        assert a <= 1.25 * b + 2e-4, (pre, sdpa)
    assert plain[0] > 2.5 * pre[0] and plain[2] > 2.5 * pre[2], (plain, pre)


def test_dit_block_backward_at_peaky_self_attention_logits_with_and_without_the_prescaled_q():
    """The block's backward (DiTBlockFn) with norm_q's weight x 8 (self-attention logits x 8) at 2048 tokens (kernel 3 in the forward)
    against fp32 autograd of the oracle: dx and the self-attention's gradients within 1.25 x the distance of the reference's own bf16
    arithmetic (the oracle's autograd on bf16 tensors: near-one-hot rows amplify every upstream rounding, 2-3e-2 for either path), with
    the pre-scaled q (shipped); ops.options(attn_q_prescale=False) — rounds 1-5's training path — is further away in sum.  (Weight
    gradients add the attention's per-token errors over all tokens, so most of the operator-level gap of the test above averages
    out here and at production size: profiles/r06/fullsize_train_parity_peaky3.json.)"""
    import gen_inputs as gi
    from goal_force_amd import ops
    from goal_force_amd.dit import DiTBlock, RopeTable, precompute_freqs_cis_3d
    from goal_force_amd.training import block_forward
    from oracle import wan_oracle as wo
    cfg = gi.MID
    sd = gi.block_sd(torch.Generator().manual_seed(21), cfg["dim"], cfg["ffn_dim"], "", BF)
    sd["self_attn.norm_q.weight"] = (sd["self_attn.norm_q.weight"].float() * 8).to(BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 2048, 512, seed=22)
    dout = torch.randn(x.shape, generator=torch.Generator().manual_seed(23)).to(BF)
    sdf = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    xf = x.float().requires_grad_(True)
    freqs = wo.rope_freqs_3d(cfg["dim"] // cfg["num_heads"], 8, 16, 16)
    wo.dit_block(xf, ctx.float(), t_mod.float(), freqs, sdf, "", cfg["num_heads"], cfg["eps"]).backward(dout.float())
    sdb = {k: v.clone().requires_grad_(True) for k, v in sd.items()}                     # the reference's bf16 arithmetic under autograd
    xb = x.clone().requires_grad_(True)
    wo.dit_block(xb, ctx, t_mod, freqs, sdb, "", cfg["num_heads"], cfg["eps"]).backward(dout)
    blk = DiTBlock(False, cfg["dim"], cfg["num_heads"], cfg["ffn_dim"], cfg["eps"])
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(BF).cuda()
    rope = RopeTable.from_grid(precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"]), 8, 16, 16, "cuda")
    names = ["self_attn.q.weight", "self_attn.k.weight", "self_attn.v.weight", "self_attn.norm_q.weight", "self_attn.o.weight"]
    named = dict(blk.named_parameters())
    err = {}
    for flag in (True, False):
        for p_ in blk.parameters():
            p_.grad = None
        xc = x[0].cuda().requires_grad_(True)
        with ops.options(attn_q_prescale=flag):
            block_forward(blk, xc, ctx[0].cuda(), t_mod.cuda(), rope).backward(dout[0].cuda())
        err[flag] = {"dx": rel_l2(xc.grad.float().cpu(), xf.grad[0])}
        err[flag].update({n: rel_l2(named[n].grad.float().cpu().reshape(sdf[n].grad.shape), sdf[n].grad) for n in names})
    ref = {"dx": rel_l2(xb.grad.float(), xf.grad)}
    ref.update({n: rel_l2(sdb[n].grad.float(), sdf[n].grad) for n in names})
    print(f"block backward, self-attention logits x 8, vs fp32 autograd: pre-scaled q {err[True]}  plain q {err[False]}  reference bf16 {ref}")
    for n in ["dx"] + names:
        assert err[True][n] <= 1.25 * ref[n], (n, err[True], ref)
    assert sum(err[True].values()) < sum(err[False].values()), err


def test_dit_block_backward_is_the_same_whatever_the_forward_kept(monkeypatch):
    """training.set_keep_level none / attn / wide: kept tensors are the forward's own values, so the gradients agree to rounding (the fused
    forward rounds x1 / x2b once where the un-fused recompute rounds per op), for a trainable and for a frozen block."""
    import gen_inputs as gi
    from goal_force_amd import training
    from goal_force_amd.dit import DiTBlock, RopeTable, precompute_freqs_cis_3d
    cfg = gi.MID
    sd = gi.block_sd(torch.Generator().manual_seed(31), cfg["dim"], cfg["ffn_dim"], "", BF)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], 960, 512, seed=32)
    dout = torch.randn(x.shape, generator=torch.Generator().manual_seed(33)).to(BF)
    blk = DiTBlock(False, cfg["dim"], cfg["num_heads"], cfg["ffn_dim"], cfg["eps"])
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(BF).cuda()
    rope = RopeTable.from_grid(precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"]), 5, 12, 16, "cuda")
    res = {}
    for frozen in (False, True):
        for p_ in blk.parameters():
            p_.requires_grad_(not frozen)
        for level in ("none", "attn", "wide"):
            monkeypatch.setattr(training, "KEEP_ATTENTION", True)      # (restores the module state after the test)
            monkeypatch.setattr(training, "KEEP_WIDE", False)
            monkeypatch.setattr(training, "KEEP_AUTO", True)
            training.set_keep_level(level)
            for p_ in blk.parameters():
                p_.grad = None
            xc = x[0].cuda().requires_grad_(True)
            with torch.enable_grad():
                training.block_forward(blk, xc, ctx[0].cuda(), t_mod.cuda(), rope).backward(dout[0].cuda())
            res[(frozen, level)] = (xc.grad.float().cpu(), {n: p_.grad.float().cpu() for n, p_ in blk.named_parameters() if p_.grad is not None})
    for frozen in (False, True):
        dx0, g0 = res[(frozen, "none")]
        assert frozen == (len(g0) == 0)
        for level in ("attn", "wide"):
            dx, g = res[(frozen, level)]
            assert rel_l2(dx, dx0) < 6e-3, (frozen, level)
            assert g.keys() == g0.keys()
            for n in g:
                assert rel_l2(g[n], g0[n]) < (6e-2 if n.endswith("attn.k.bias") else 1e-2), (frozen, level, n)
        assert torch.equal(res[(frozen, "attn")][0], dx0), "keeping only the attention output changes no bit"


def _tiny_train_models():
    import gen_inputs as gi
    from goal_force_amd.controlnet import ControlNet
    from goal_force_amd.dit import WanModel
    cfg = gi.TINY
    dit = WanModel(has_image_input=False, require_clip_embedding=False, **cfg)
    dit.load_state_dict(gi.dit_sd(cfg, seed=41), strict=True)
    cn = ControlNet(gi.TINY_CONTROLNET_LAYERS, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"])
    cn.load_state_dict(gi.controlnet_sd(cfg, gi.TINY_CONTROLNET_LAYERS, seed=42), strict=True)
    dit, cn = dit.to(BF).cuda(), cn.to(BF).cuda()
    for p_ in dit.parameters():
        p_.requires_grad_(False)
    return dit, cn


def _tiny_train_pipe(dit, cn):
    from goal_force_amd.pipeline import WanVideoPipeline
    pipe = WanVideoPipeline.from_modules(dit, None, cn, None, device="cuda")
    pipe.scheduler.set_timesteps(1000, training=True)                  # utils.py:560
    return pipe


def test_training_step_vs_reference_golden():
    """loss and every ControlNet gradient of one training_loss call against the reference's own run (g9_training.npz):
    bar = 2x the reference-bf16's own distance from its fp32 run, floor 2e-2."""
    import os
    import numpy as np
    import gen_inputs as gi
    from conftest import GOLDEN
    from goal_force_amd import training as tr
    g = np.load(os.path.join(GOLDEN, "g9_training.npz"))
    dit, cn = _tiny_train_models()
    pipe = _tiny_train_pipe(dit, cn)
    inp = {k: v.cuda() for k, v in gi.train_inputs().items()}
    with torch.enable_grad():      # through the pipeline method, as the reference's training module calls it (GF:180)
        loss = pipe.training_loss(input_latents=inp["input_latents"], noise=inp["noise"], context=inp["context"],
                                  y=inp["y"], control_signal_video_latents=inp["control"], timestep_id=gi.TRAIN_TIMESTEP_ID)
        loss.backward()
    lf, lb = float(g["loss_f32"]), float(g["loss_bf16"])
    lv = float(loss.detach())
    assert abs(lv - lf) <= 2 * abs(lb - lf) + 1e-2 * lf, f"loss {lv} vs fp32 {lf} (reference bf16 {lb})"
    named = dict(cn.named_parameters())
    names = [str(n) for n in g["names"]]
    assert sorted(named) == names
    assert all(p_.grad is None for p_ in dit.parameters())
    for i, n in enumerate(names):
        gr = named[n].grad
        assert gr is not None and gr.dtype == BF and gr.shape == named[n].shape, n
        gf = gr.double().cpu().flatten()
        idx = gi.grad_sample_index(gf.numel(), seed=2000 + i)
        s32, s16 = torch.from_numpy(g["sample_f32"][i]), torch.from_numpy(g["sample_bf16"][i])
        e_ref = float((s16 - s32).norm() / s32.norm())
        e = float((gf[idx] - s32).norm() / s32.norm())
        assert e <= 2 * e_ref + 2e-2, f"{n}: sample rel err {e:.3e} (reference bf16 {e_ref:.3e})"
        nrm = float(g["norm_f32"][i])
        assert abs(float(gf.norm()) - nrm) <= (2 * abs(float(g["norm_bf16"][i]) - nrm) + 3e-2 * nrm), n


def test_training_steps_reduce_the_loss():
    """launch_training_task's inner loop (utils.py:797-812): zero_grad, loss, backward, clip, AdamW step."""
    import gen_inputs as gi
    from goal_force_amd import training as tr
    dit, cn = _tiny_train_models()
    pipe = _tiny_train_pipe(dit, cn)
    inp = {k: v.cuda() for k, v in gi.train_inputs().items()}
    opt = tr.AdamW(cn.parameters(), lr=2e-4, weight_decay=1e-2)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        with torch.enable_grad():
            loss = tr.training_loss(pipe, input_latents=inp["input_latents"], noise=inp["noise"], context=inp["context"],
                                    y=inp["y"], control_signal_video_latents=inp["control"], timestep_id=gi.TRAIN_TIMESTEP_ID)
            loss.backward()
        opt.step(max_grad_norm=1.0)
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    want = math.sqrt(sum(float(p_.grad.float().pow(2).sum()) for p_ in cn.parameters() if p_.grad is not None))
    assert abs(opt.grad_norm() - want) < 1e-4 * want
    sd = tr.controlnet_state_dict(cn)
    assert all(k.startswith("pipe.controlnet.") for k in sd) and len(sd) == len(cn.state_dict())


def test_derived_weight_caches_follow_optimizer_steps():
    """train -> validate -> train -> validate in one process: the K-padded patch-embedding copy, the
    all-zero flag of the ControlNet and fp8 weight copies must be rebuilt after AdamW steps, so model_fn on the trained
    module equals model_fn on a freshly constructed module holding the same weights, bit for bit."""
    import gen_inputs as gi
    from goal_force_amd import training as tr
    from goal_force_amd._lib import GoalForceError
    from goal_force_amd.controlnet import ControlNet
    from goal_force_amd.dit import enable_fp8
    from goal_force_amd.model_fn import model_fn_wan_video
    cfg = gi.TINY
    dit, cn = _tiny_train_models()
    with torch.no_grad():
        for c in cn.controlnet_zero_convs_after:          # a never-trained ControlNet: zero-convs exactly zero
            c.weight.zero_()
            c.bias.zero_()
    pipe = _tiny_train_pipe(dit, cn)
    inp = {k: v.cuda() for k, v in gi.train_inputs().items()}
    ts = torch.tensor([500.0], dtype=BF).cuda()

    def validate(module):
        with torch.no_grad():
            return model_fn_wan_video(dit, latents=inp["noise"], timestep=ts, context=inp["context"], y=inp["y"],
                                      controlnet=module, control_signal_video_latents=inp["control"])

    before = validate(cn)                                 # fills the caches with the pre-training weights
    assert cn.all_zero()
    opt = tr.AdamW(cn.parameters(), lr=1e-3, weight_decay=0.0)
    for _ in range(2):
        opt.zero_grad()
        with torch.enable_grad():
            loss = tr.training_loss(pipe, input_latents=inp["input_latents"], noise=inp["noise"], context=inp["context"],
                                    y=inp["y"], control_signal_video_latents=inp["control"], timestep_id=gi.TRAIN_TIMESTEP_ID)
            loss.backward()
        opt.step()
    assert not cn.all_zero(), "the zero-convs were trained: the ControlNet may no longer be skipped"
    after = validate(cn)
    fresh = ControlNet(cn.num_layers, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"])
    fresh.load_state_dict({k: v.detach().cpu() for k, v in cn.state_dict().items()}, strict=True)
    fresh = fresh.to(BF).cuda()
    assert torch.equal(after, validate(fresh)) and not torch.equal(after, before)
    # fp8 copies: refreshed after an update as well, and training through an fp8 block is refused
    enable_fp8(cn)
    a8 = validate(cn)
    opt.zero_grad()
    with pytest.raises(GoalForceError, match="enable_fp8"):
        with torch.enable_grad():
            tr.training_loss(pipe, input_latents=inp["input_latents"], noise=inp["noise"], context=inp["context"],
                             y=inp["y"], control_signal_video_latents=inp["control"], timestep_id=gi.TRAIN_TIMESTEP_ID)
    with torch.no_grad():
        for p_ in cn.controlnet_dit.parameters():
            if p_.dim() == 2:
                p_.mul_(0.5)                               # an in-place update torch sees (version bump)
    b8 = validate(cn)
    enable_fp8(fresh)
    with torch.no_grad():
        for p_ in fresh.controlnet_dit.parameters():
            if p_.dim() == 2:
                p_.mul_(0.5)
    enable_fp8(fresh)                                      # fresh copies cast from the updated masters
    assert torch.equal(b8, validate(fresh)) and not torch.equal(a8, b8)


def test_training_loop_vs_the_reference_launch_training_task(tmp_path):
    """`training.launch_training_task` against g17 = the reference's OWN `launch_training_task` (utils.py:734-826: Accelerate,
    torch AdamW, ConstantLR, clip_grad_norm_, ModelLogger) around its own `WanTrainingModule.forward`, run on CPU on the same tiny
    pipeline, the same four 832x480x5-frame items in the recorded order, the same per-step random draws (gen_inputs.train_loop_draws):
      * bookkeeping bit for bit: learning rate of every step (lr / 3 for five steps, then lr), checkpoint names (step-3, step-6 and
        the final step-8), checkpoint keys, the bf16-rounded timestep of every step;
      * per-step loss and pre-clip gradient norm: no further from the reference's fp32 run than its bf16 run is (x 2, small floor);
      * the parameter change of the final checkpoint (sampled entries of every tensor): the same bar."""
    import argparse
    import datetime
    import os
    import numpy as np
    import gen_inputs as gi
    from conftest import GOLDEN
    from safetensors.torch import load_file
    from goal_force_amd import training as tr
    from goal_force_amd.vae import WanVideoVAE
    g = np.load(os.path.join(GOLDEN, "g17_training_loop.npz"))
    cfg = gi.TRAIN_LOOP
    items = gi.training_items()
    assert gi.same_checksum(gi.checksum([torch.from_numpy(np.stack([np.array(f) for f in it["video"]])).float() for it in items]
                                        + [it["control_video"] for it in items]), g["ck_items"])
    dit, cn = _tiny_train_models()
    pipe = _tiny_train_pipe(dit, cn)
    g6 = np.load(os.path.join(GOLDEN, "g6_vae.npz"))
    vae = WanVideoVAE()
    vae.load_state_dict({"model." + k: t for k, t in gi.vae_decoder_sd(list(g6["names"]), g6["shapes"], seed=61).items()}, strict=True)
    pipe.vae = vae.to(BF).cuda()
    pipe._after_models_attached()
    for p_ in pipe.vae.parameters():
        p_.requires_grad_(False)
    inp = gi.tiny_inputs()

    class Prompter:
        def encode_prompt(self, prompt, positive=True, device="cuda"):
            return (inp["ctx_posi"] if prompt == gi.PIPELINE_PROMPTS[0] else inp["ctx_nega"]).to(device)
    pipe.prompter = Prompter()
    order = [int(v) for v in g["order_bf16"]]
    assert order == [int(v) for v in g["order_f32"]] and sorted(order[:4]) == sorted(order[4:]) == [0, 1, 2, 3]

    class Replay(torch.utils.data.Dataset):              # the recorded shuffle of the reference's DataLoader, epoch after epoch
        calls = 0

        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            it = items[order[Replay.calls]]
            Replay.calls += 1
            return dict(it)
    rec = {"loss": [], "timestep": []}

    def forward(pipe_, data):
        k = len(rec["loss"])
        noise, tid = gi.train_loop_draws(k, hi=int(cfg["max_timestep_boundary"] * 1000))
        inputs = tr.forward_preprocess(pipe_, data)
        assert tuple(inputs["noise"].shape) == tuple(noise.shape)
        inputs["noise"] = noise.to(BF).cuda()             # generate_noise: fp32 draws rounded to the pipeline's dtype (UTIL:117-122)
        rec["timestep"].append(float(pipe_.scheduler.timesteps[tid].to(BF).float()))
        with torch.enable_grad():
            loss = tr.training_loss(pipe_, **inputs, timestep_id=tid)
        rec["loss"].append(float(loss.detach()))
        return loss
    logged = []
    args = argparse.Namespace(learning_rate=cfg["learning_rate"], weight_decay=cfg["weight_decay"], dataset_num_workers=0, save_steps=cfg["save_steps"],
                              num_epochs=cfg["num_epochs"], gradient_accumulation_steps=1, find_unused_parameters=False, controlnet_checkpoint=None,
                              output_path=str(tmp_path), remove_prefix_in_ckpt=cfg["remove_prefix_in_ckpt"], control_signal_type=cfg["control_signal_type"],
                              num_frames=cfg["num_frames"], max_grad_norm=cfg["max_grad_norm"])
    lrs, norms = [], []
    real_end = tr.ModelLogger.on_step_end

    def spy_end(self, pipe_, loss, learning_rate, grad_norm=None, save_steps=None):
        lrs.append(learning_rate)
        norms.append(grad_norm)
        return real_end(self, pipe_, loss, learning_rate, grad_norm=grad_norm, save_steps=save_steps)
    tr.ModelLogger.on_step_end = spy_end
    try:
        logger = tr.launch_training_task(Replay(), pipe, args=args, forward=forward, shuffle=False, log=lambda r, s_: logged.append((s_, r)),
                                         now=datetime.datetime(2026, 1, 2, 3, 4, 5))
    finally:
        tr.ModelLogger.on_step_end = real_end
    run_dir = os.path.join(str(tmp_path), "2026-01-02_03-04-05")
    files = sorted(os.listdir(run_dir), key=lambda f: int(f.split("-")[1].split(".")[0]))
    assert files == [str(f) for f in g["files"]] == ["step-3.safetensors", "step-6.safetensors", "step-8.safetensors"] and logger.num_steps == 8
    assert lrs == [float(v) for v in g["lr_bf16"]], (lrs, g["lr_bf16"])                      # lr / 3 x 5, then lr: the same floats
    assert rec["timestep"] == [float(v) for v in g["timestep_bf16"]]
    lf, lb = g["loss_f32"], g["loss_bf16"]
    for k in range(8):
        assert abs(rec["loss"][k] - lf[k]) <= 2 * abs(lb[k] - lf[k]) + 2e-2 * abs(lf[k]), (k, rec["loss"][k], lf[k], lb[k])
        nf, nb = g["grad_norm_f32"][k], g["grad_norm_bf16"][k]
        assert abs(norms[k] - nf) <= 2 * abs(nb - nf) + 5e-2 * nf, (k, norms[k], nf, nb)
    sd = load_file(os.path.join(run_dir, files[-1]))
    names = [str(n) for n in g["ckpt_keys"]]
    assert sorted(sd) == names and all(v.dtype == BF for v in sd.values())
    csd0 = gi.controlnet_sd(gi.TINY, gi.TINY_CONTROLNET_LAYERS, seed=42)
    got, s32, s16 = [], [], []
    for i, n in enumerate(names):
        d = (sd[n].double() - csd0[n[len("pipe.controlnet."):]].to(BF).double()).flatten()
        got.append(d[gi.grad_sample_index(d.numel(), seed=4000 + i, k=2048)])
        s32.append(torch.from_numpy(g["delta_sample_f32"][i]))
        s16.append(torch.from_numpy(g["delta_sample_bf16"][i]))
    got, s32, s16 = torch.cat(got), torch.cat(s32), torch.cat(s16)
    e, e_ref = float((got - s32).norm() / s32.norm()), float((s16 - s32).norm() / s32.norm())
    print(f"g17: parameter change after 8 steps vs the reference's fp32 run: {e:.3e} (the reference's bf16 run: {e_ref:.3e}); "
          f"losses {[round(v, 4) for v in rec['loss']]}")
    assert e <= 1.5 * e_ref + 2e-2, f"parameter change: {e:.3e} vs fp32 (reference bf16 {e_ref:.3e})"
    assert float(got.abs().max()) > 0 and logged == [], "8 steps: the every-10-steps log never fires"
    # resume rule (utils.py:773-785): output next to the checkpoint, LR schedule fast-forwarded, step count continues at N + 1 (sic)
    Replay.calls = 0
    rec["loss"].clear()
    rec["timestep"].clear()
    lrs.clear()
    args.controlnet_checkpoint, args.num_epochs, args.save_steps = os.path.join(run_dir, "step-6.safetensors"), 1, 100
    tr.ModelLogger.on_step_end = spy_end
    try:
        logger2 = tr.launch_training_task(Replay(), pipe, args=args, forward=forward, shuffle=False)
    finally:
        tr.ModelLogger.on_step_end = real_end
    assert logger2.output_path == run_dir and logger2.num_steps == 6 + 1 + 4 and lrs == [cfg["learning_rate"] * (1.0 / 3) * 3.0] * 4
    assert os.path.exists(os.path.join(run_dir, "step-11.safetensors"))


def _loop_worker(rank, world, port, out):
    """Two ranks (gloo rendezvous, both on cuda:0 — the box has one GPU) through launch_training_task with a toy trainable module."""
    import argparse
    import os
    import types
    import torch.distributed as dist
    from PIL import Image
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from goal_force_amd import training as tr
        torch.manual_seed(5)
        cn = torch.nn.Module()
        cn.w = torch.nn.Parameter(torch.randn(64, 32).to(BF).cuda())
        cn.b = torch.nn.Parameter(torch.zeros(32).to(BF).cuda())
        pipe = types.SimpleNamespace(controlnet=cn, device=torch.device("cuda", 0))
        frame = Image.new("RGB", (832, 480))
        good = lambda k: {"video": [frame] * 5, "control_video": torch.zeros((5, 480, 832, 3), dtype=BF), "file_id": f"item{k}", "k": float(k + 1)}

        class DS(torch.utils.data.Dataset):          # 5 items: DistributedSampler pads to 3 per rank; item 3 is a clip that failed to load
            def __len__(self):
                return 5

            def __getitem__(self, i):
                return None if i == 3 else good(i)
        seen = []

        def forward(pipe_, data):
            seen.append(data["file_id"])
            x = torch.full((64,), data["k"], device="cuda")
            return ((x @ pipe_.controlnet.w.float()) + pipe_.controlnet.b.float()).pow(2).mean() * 1e-3
        lrs = []
        real_end = tr.ModelLogger.on_step_end

        def spy_end(self, pipe_, loss, learning_rate, grad_norm=None, save_steps=None):
            lrs.append(learning_rate)
            return real_end(self, pipe_, loss, learning_rate, grad_norm=grad_norm, save_steps=save_steps)
        tr.ModelLogger.on_step_end = spy_end
        args = argparse.Namespace(learning_rate=3e-3, weight_decay=1e-2, dataset_num_workers=0, save_steps=2, num_epochs=2, gradient_accumulation_steps=1,
                                  find_unused_parameters=False, controlnet_checkpoint=None, output_path=os.path.join(out, "run"), remove_prefix_in_ckpt=None,
                                  control_signal_type="direct_force_and_goal_force_and_mass", num_frames=5, max_grad_norm=1.0)
        import datetime
        logger = tr.launch_training_task(DS(), pipe, args=args, forward=forward, now=datetime.datetime(2026, 1, 1))
        dist.barrier()               # rank 0 may still be writing the last checkpoint when rank 1 comes back
        torch.save({"w": cn.w.detach().cpu(), "b": cn.b.detach().cpu(), "seen": seen, "lrs": lrs, "steps": logger.num_steps,
                    "files": sorted(os.listdir(logger.output_path)) if os.path.isdir(logger.output_path) else []}, os.path.join(out, f"loop{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_training_loop_on_two_ranks_shards_the_data_averages_gradients_and_skips_bad_batches_together(tmp_path):
    """launch_training_task under torch.distributed (what the reference gets from accelerator.prepare + DDP): each rank draws its share of
    one shuffled order (padded to equal length), gradients are averaged, so both ranks hold the same parameters after every step; a clip
    that failed to load on ONE rank makes BOTH skip that step (the all-reduce consensus of utils.py:682-698); the LR schedule advances
    world-size steps per training step (Accelerate's prepared scheduler); rank 0 alone writes the checkpoints."""
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(tmp_path, f"loop{r}.pt")) for r in (0, 1))
    assert torch.equal(a["w"], b["w"]) and torch.equal(a["b"], b["b"]), "the ranks trained the same parameters"
    assert a["steps"] == b["steps"] and len(a["seen"]) == len(b["seen"]) == a["steps"]
    # 5 items padded to 6 = 3 per rank and epoch; the rank that draws the failed clip drags its partner's step along: 2 epochs x 3 - skipped
    # (DistributedSampler, seed 0: epoch 0 draws [4, 1, 2] / [0, 3, 4], epoch 1 [0, 2, 1] / [4, 3, 0]: the second iteration of each epoch is skipped)
    assert a["steps"] == 4 and a["seen"] == ["item4", "item2", "item0", "item1"] and b["seen"] == ["item0", "item4", "item4", "item0"], (a["seen"], b["seen"])
    assert "item3" not in a["seen"] + b["seen"]
    lr, third = 3e-3, 3e-3 * (1.0 / 3)
    want = [third, third] + [third * 3.0] * (a["steps"] - 2)         # ConstantLR's 5 iterations are used up after ceil(5 / 2) = 3 training steps
    assert a["lrs"][:2] == want[:2] and a["lrs"][3:] == want[3:] and a["lrs"][2] == third, a["lrs"]
    assert a["files"] == b["files"] == ["step-2.safetensors", "step-4.safetensors"]
