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


@pytest.mark.parametrize("sq,skv,heads", [(72, 72, 2), (300, 200, 3), (1000, 512, 4), (128, 64, 1), (257, 130, 2)])
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
