"""umT5 text encoder: CPU oracle pinned to the reference WanTextEncoder; HIP encoder vs the same goldens."""
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN, rel_l2
from oracle import t5_oracle as to

BF = torch.bfloat16
torch.set_grad_enabled(False)


def _fix():
    g = np.load(os.path.join(GOLDEN, "g8_text_encoder.npz"))
    sd = gi.t5_sd(gi.T5_TINY, seed=81)
    assert gi.same_checksum(gi.checksum(sd), g["ck_weights"])
    return g, sd, torch.from_numpy(g["ids"]), torch.from_numpy(g["mask"])


def test_oracle_matches_reference():
    g, sd, ids, mask = _fix()
    cfg = gi.T5_TINY
    assert torch.equal(to.encode(ids, mask, sd, cfg["num_heads"], cfg["num_layers"]), gi.from_u16(g["encode_bf16"]))
    assert torch.equal(to.encode_prompt_ids(ids, mask, sd, cfg["num_heads"], cfg["num_layers"]),
                       gi.from_u16(g["prompt_emb_bf16"]))
    sd32 = {k: v.float() for k, v in sd.items()}
    assert rel_l2(to.encode(ids, mask, sd32, cfg["num_heads"], cfg["num_layers"]), torch.from_numpy(g["encode_f32"])) < 1e-5


def test_state_dict_names_match_reference():
    from goal_force_amd.text_encoder import WanTextEncoder
    cfg = gi.T5_TINY
    m = WanTextEncoder(**cfg)
    sd = gi.t5_sd(cfg, seed=81)   # these keys were loaded strict=True into the reference class
    assert sorted(m.state_dict().keys()) == sorted(sd.keys())
    m.load_state_dict(sd, strict=True)
    # bucket table equals the oracle's restatement of T5:165-190
    b = m.blocks[0].pos_embedding(24, 24)
    assert torch.equal(b[None], to.rel_bias(sd["blocks.0.pos_embedding.embedding.weight"].float(), 24, 24))


@pytest.mark.gpu
def test_hip_text_encoder_vs_reference_golden():
    from goal_force_amd.text_encoder import WanPrompter, WanTextEncoder
    g, sd, ids, mask = _fix()
    enc = WanTextEncoder(**gi.T5_TINY)
    enc.load_state_dict(sd, strict=True)
    enc = enc.to(BF).cuda()
    got = enc(ids.cuda(), mask.cuda()).cpu()
    f32 = torch.from_numpy(g["encode_f32"])
    ref_bf = gi.from_u16(g["encode_bf16"]).float()
    nv = int(mask.sum())
    # rows past the prompt length are zeroed by the prompter; compare the real tokens
    e, e_ref = rel_l2(got.float()[:, :nv], f32[:, :nv]), rel_l2(ref_bf[:, :nv], f32[:, :nv])
    assert e < max(5e-3, 1.5 * e_ref), f"vs fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
    pr = WanPrompter()
    pr.fetch_models(enc)
    emb = pr.encode_ids(ids, mask).cpu()
    assert float(emb[:, nv:].abs().sum()) == 0
    assert rel_l2(emb.float(), torch.from_numpy(g["prompt_emb_f32"])) < max(5e-3, 1.5 * e_ref)
    # no mask == all keys valid
    full = enc(ids.cuda()).cpu()
    ref_full = to.encode(ids, None, {k: v.float() for k, v in sd.items()}, 4, 2)
    e_ref_full = rel_l2(to.encode(ids, None, sd, 4, 2).float(), ref_full)   # the bf16 arithmetic's own noise (unscaled logits)
    e_full = rel_l2(full.float(), ref_full)
    assert e_full < max(5e-3, 1.5 * e_ref_full), f"unmasked: vs fp32 {e_full:.3e} (bf16 oracle {e_ref_full:.3e})"


@pytest.mark.gpu
def test_gemm_mul_epilogue_and_biased_softmax():
    import torch.nn.functional as F
    from goal_force_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn((300, 256), generator=g).to(BF)
    w = (torch.randn((512, 256), generator=g) / 16).to(BF)
    r = torch.randn((300, 512), generator=g).to(BF)
    got = ops.gemm(x.cuda(), w.cuda(), epilogue=ops.EPI_BIAS_MUL, resid=r.cuda()).cpu()
    ref = F.linear(x.float(), w.float()).to(BF) * r
    assert rel_l2(got.float(), ref.float()) < 2e-3
    s = (torch.randn((40, 48), generator=g) * 3).to(BF)
    b = torch.randn((40, 48), generator=g).to(BF)
    p = ops.softmax_rows(s.cuda(), 1.0, 64, bias=b.cuda(), nvalid=33).cpu()
    refp = torch.softmax((s + b)[:, :33].float(), -1)
    assert rel_l2(p[:, :33].float(), refp) < 3e-3 and float(p[:, 33:].abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,M,N,K,heads_layout", [(64, 512, 512, 64, True), (5, 300, 520, 128, False), (3, 40, 64, 448, False),
                                                  (21, 1560, 1560, 384, False), (7, 512, 64, 512, True), (1, 17, 8, 64, False)])
def test_batched_gemm_and_transpose_equal_the_per_problem_loop(B, M, N, K, heads_layout):
    """gf_gemm_bf16_batched / gf_transpose_pad_batched (one launch for all heads of the umT5 attention, all frames of the VAE attention)
    against one gf_gemm_bf16 / gf_transpose_pad call per problem: bit for bit, with the operands as strided head views of one
    [rows, B*d] tensor or as separate matrices, ragged M / N included; argument errors.  The per-problem calls are pinned to the 8-wave
    kernel (options(prefer_8wave=1)), which the batched launch uses: for M >= 512 gf_gemm_bf16 would take the 4-wave kernel, whose column tiles
    start their K loops at rotated K tiles — the same products summed in another order."""
    from goal_force_amd import ops
    from goal_force_amd._lib import GoalForceError
    g = torch.Generator().manual_seed(B * 1000 + M)
    if heads_layout:      # heads side by side: a [M, B*K] tensor viewed as [B, M, K]
        a = torch.randn((M, B * K), generator=g).to(BF).cuda().view(M, B, K).permute(1, 0, 2)
        w = torch.randn((N, B * K), generator=g).to(BF).cuda().view(N, B, K).permute(1, 0, 2)
        out = torch.zeros((M, B * N), dtype=BF).cuda().view(M, B, N).permute(1, 0, 2)
    else:
        a = torch.randn((B, M, K), generator=g).to(BF).cuda()
        w = torch.randn((B, N, K), generator=g).to(BF).cuda()
        out = None
    got = ops.gemm_batched(a, w, out=out)
    assert tuple(got.shape) == (B, M, N)
    with ops.options(prefer_8wave=1):
        for b in range(B):
            ref = ops.gemm(a[b], w[b])
            assert torch.equal(got[b], ref), (b, int((got[b] != ref).sum()))
    ref4 = ops.gemm(a[B - 1], w[B - 1])                          # whatever kernel the single call takes: the same numbers up to fp32 summation order
    assert rel_l2(got[B - 1].float(), ref4.float()) < 1e-3
    assert float(got.float().abs().max()) > 1
    rpad = -(-M // 64) * 64
    tb = ops.transpose_pad_batched(a, rpad)
    for b in range(B):
        assert torch.equal(tb[b], ops.transpose_pad(a[b], rpad))
    with pytest.raises(GoalForceError):
        ops.gemm_batched(a, w[:, :, : K - 8])                     # K mismatch
    with pytest.raises(GoalForceError):
        ops.gemm_batched(a[:, :, :32], w[:, :, :32])              # K not a multiple of 64


@pytest.mark.gpu
def test_hip_umt5_xxl_full_size_within_reference_drift():
    """umT5-XXL at its real size (24 layers, dim 4096, 64 heads, FFN 10240, 512 tokens of which 40 are prompt: what the pipeline runs
    twice per video) against the reference's arithmetic (oracle/t5_oracle.py on this GPU: bf16 = what the reference computes, fp32 =
    the yardstick).  Random-init weights, q projections x 0.2 (logit std ~ 2.6; unscaled, a random 24-layer T5 is chaotic for every
    bf16 arithmetic — tests/fullsize_t5_parity.py).  Bar as for the DiT at full size: HIP-vs-fp32 <= 1.25 x (reference
    bf16 vs fp32), and the reference's own drift small enough for the comparison to mean something."""
    import fullsize_t5_parity as ft
    rep = ft.run(valid=(40,), q_scale=0.2, log=lambda s: print(s, flush=True))
    assert rep["config"]["layers"] == 24 and rep["config"]["params"] > 5.5e9
    r = rep["valid_tokens"]["40"]
    assert r["ref_bf16_vs_fp32"] < 0.2, r
    assert r["hip_bf16_vs_fp32"] <= 1.25 * r["ref_bf16_vs_fp32"], r
