"""umT5-XXL text encoder (WanTextEncoder) + WanPrompter — SURVEY.md §8f rank 2, the producer of `context`
[1,512,4096] for the denoising loop (GF:808-822).  Mirrors diffsynth/models/wan_video_text_encoder.py ("T5") and
diffsynth/prompters/wan_prompter.py ("PR"): same constructor arguments, state_dict keys
(`token_embedding.weight`, `blocks.N.{norm1,norm2}.weight`, `blocks.N.attn.{q,k,v,o}.weight`,
`blocks.N.ffn.{gate.0,fc1,fc2}.weight`, `blocks.N.pos_embedding.embedding.weight`, `norm.weight`) and forward
`(ids [1,L], mask [1,L]) -> [1,L,dim]`.

All arithmetic runs on the HIP kernels: T5LayerNorm = gf_rmsnorm_rope without RoPE (same fp32-norm / bf16-weight
rounding, T5:22-35); q/k/v/o and the GEGLU FFN on the MFMA GEMM (gate with the GELU-tanh epilogue, fc1 with the
multiply epilogue, fc2/o with the residual epilogue); attention per head (head_dim 64, no 1/sqrt(d) scaling,
T5:78-83) as GEMM -> gf_softmax_rows(+ relative position bias, key mask) -> GEMM.  Two prompts of 512 tokens per
video: this path is launch-bound, not a throughput kernel.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from ._lib import GoalForceError


class T5LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x2):
        y = x2.clone()
        ops.rmsnorm_rope(y, self.weight, None, None, head_dim=8, eps=self.eps)
        return y


class T5RelativeEmbedding(nn.Module):
    """T5:147-190 — bucketed relative position bias, bidirectional, max_dist 128."""

    def __init__(self, num_buckets, num_heads, bidirectional, max_dist=128):
        super().__init__()
        self.num_buckets, self.num_heads, self.bidirectional, self.max_dist = num_buckets, num_heads, bidirectional, max_dist
        self.embedding = nn.Embedding(num_buckets, num_heads)

    def buckets(self, lq, lk, device):
        rel_pos = torch.arange(lk, device=device).unsqueeze(0) - torch.arange(lq, device=device).unsqueeze(1)
        if self.bidirectional:
            nb = self.num_buckets // 2
            rel_buckets = (rel_pos > 0).long() * nb
            rel_pos = torch.abs(rel_pos)
        else:
            nb = self.num_buckets
            rel_buckets = 0
            rel_pos = -torch.min(rel_pos, torch.zeros_like(rel_pos))
        max_exact = nb // 2
        large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(self.max_dist / max_exact)
                             * (nb - max_exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        return rel_buckets + torch.where(rel_pos < max_exact, rel_pos, large)

    def forward(self, lq, lk):
        """-> [N, lq, lk] bias in the embedding dtype (a table lookup: data movement only)."""
        b = self.buckets(lq, lk, self.embedding.weight.device)
        return self.embedding.weight[b].permute(2, 0, 1).contiguous()


class T5Attention(nn.Module):
    def __init__(self, dim, dim_attn, num_heads, dropout=0.1):
        super().__init__()
        assert dim_attn % num_heads == 0
        self.dim, self.dim_attn, self.num_heads, self.head_dim = dim, dim_attn, num_heads, dim_attn // num_heads
        self.q = nn.Linear(dim, dim_attn, bias=False)
        self.k = nn.Linear(dim, dim_attn, bias=False)
        self.v = nn.Linear(dim, dim_attn, bias=False)
        self.o = nn.Linear(dim_attn, dim, bias=False)

    def forward(self, x2, resid, nvalid, pos_bias):
        """x2 [L, dim] (normed), resid [L, dim]; returns resid + o(attention) (T5:61-93, 140)."""
        L = x2.shape[0]
        c, n = self.head_dim, self.num_heads
        if c % 64 or L % 8:
            raise GoalForceError("T5Attention: head_dim must be a multiple of 64 and L of 8")
        q = ops.gemm(x2, self.q.weight)
        k = ops.gemm(x2, self.k.weight)
        v = ops.gemm(x2, self.v.weight)
        kp = -(-L // 64) * 64
        ctx = torch.empty((L, n * c), dtype=x2.dtype, device=x2.device)
        heads = lambda t: t.view(L, n, c).permute(1, 0, 2)                                    # [n, L, c] views: head h = columns h c ..
        # all heads per launch (one launch per head and product was 256 launches per layer, each filling 1/64 of the chip: 72 -> 23 ms per
        # prompt); the 8-wave GEMM kernel, bit-identical to per-head calls on it (tests/test_text_encoder.py)
        scores = ops.gemm_batched(heads(q), heads(k))                                         # q k^T, no scaling  [n, L, L]
        bias = None if pos_bias is None else pos_bias.reshape(n * L, L)
        p = ops.softmax_rows(scores.view(n * L, L), 1.0, kp, bias=bias, nvalid=nvalid)        # [n L, kp]
        vt = ops.transpose_pad_batched(heads(v), kp)                                          # [n, c, kp]
        ops.gemm_batched(p.view(n, L, kp), vt, out=heads(ctx))
        return ops.gemm(ctx, self.o.weight, epilogue=ops.EPI_BIAS_RESID, resid=resid)


class T5FeedForward(nn.Module):
    def __init__(self, dim, dim_ffn, dropout=0.1):
        super().__init__()
        self.dim, self.dim_ffn = dim, dim_ffn
        self.gate = nn.Sequential(nn.Linear(dim, dim_ffn, bias=False), nn.Identity())   # keys: gate.0.weight (T5:100)
        self.fc1 = nn.Linear(dim, dim_ffn, bias=False)
        self.fc2 = nn.Linear(dim_ffn, dim, bias=False)

    def forward(self, x2, resid):
        """resid + fc2(fc1(x) * gelu_tanh(gate(x)))  (T5:105-110, 141)."""
        g = ops.gemm(x2, self.gate[0].weight, epilogue=ops.EPI_BIAS_GELU_TANH)
        hmid = ops.gemm(x2, self.fc1.weight, epilogue=ops.EPI_BIAS_MUL, resid=g)
        return ops.gemm(hmid, self.fc2.weight, epilogue=ops.EPI_BIAS_RESID, resid=resid)


class T5SelfAttention(nn.Module):
    def __init__(self, dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos=True, dropout=0.1):
        super().__init__()
        self.shared_pos = shared_pos
        self.norm1 = T5LayerNorm(dim)
        self.attn = T5Attention(dim, dim_attn, num_heads, dropout)
        self.norm2 = T5LayerNorm(dim)
        self.ffn = T5FeedForward(dim, dim_ffn, dropout)
        self.pos_embedding = None if shared_pos else T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True)

    def forward(self, x2, nvalid, pos_bias=None):
        e = pos_bias if self.shared_pos else self.pos_embedding(x2.shape[0], x2.shape[0])
        x2 = self.attn(self.norm1(x2), x2, nvalid, e)
        return self.ffn(self.norm2(x2), x2)


class WanTextEncoder(nn.Module):
    """T5:209-255 (umT5-XXL encoder: vocab 256384, dim 4096, 64 heads x 64, FFN 10240, 24 layers, per-layer
    position bias)."""

    def __init__(self, vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24,
                 num_buckets=32, shared_pos=False, dropout=0.1):
        super().__init__()
        self.dim, self.dim_attn, self.dim_ffn = dim, dim_attn, dim_ffn
        self.num_heads, self.num_layers, self.num_buckets, self.shared_pos = num_heads, num_layers, num_buckets, shared_pos
        self.token_embedding = vocab if isinstance(vocab, nn.Embedding) else nn.Embedding(vocab, dim)
        self.pos_embedding = T5RelativeEmbedding(num_buckets, num_heads, bidirectional=True) if shared_pos else None
        self.blocks = nn.ModuleList([T5SelfAttention(dim, dim_attn, dim_ffn, num_heads, num_buckets, shared_pos, dropout)
                                     for _ in range(num_layers)])
        self.norm = T5LayerNorm(dim)

    @torch.no_grad()
    def forward(self, ids, mask=None):
        if ids.shape[0] != 1:
            return torch.cat([self.forward(ids[i:i + 1], None if mask is None else mask[i:i + 1])
                              for i in range(ids.shape[0])])
        if not ids.is_cuda:
            raise GoalForceError("WanTextEncoder: ids must be on the GPU (no CPU fallback exists)")
        L = ids.shape[1]
        nvalid = L
        if mask is not None:
            m = mask[0].to(torch.bool)
            nvalid = int(m.sum())
            if nvalid == 0 or not bool(m[:nvalid].all()):
                raise NotImplementedError("only prefix key masks (right-padded prompts, PR:52-57) are supported")
        x2 = self.token_embedding.weight[ids[0]].contiguous()          # embedding lookup (gather)
        e = self.pos_embedding(L, L) if self.shared_pos else None
        for blk in self.blocks:
            x2 = blk(x2, nvalid, e)
        return self.norm(x2).unsqueeze(0)


class WanPrompter:
    """PR:84-109 — tokenizer wrapper + zeroing of the embeddings past the prompt length."""

    def __init__(self, tokenizer_path=None, text_len=512):
        self.text_len = text_len
        self.text_encoder = None
        self.tokenizer = None
        self.fetch_tokenizer(tokenizer_path)

    def fetch_tokenizer(self, tokenizer_path=None):
        if tokenizer_path is not None:
            from transformers import AutoTokenizer   # local files only: there is no network here
            self.tokenizer = AutoTokenizer.from_pretrained(tokenizer_path, local_files_only=True)

    def fetch_models(self, text_encoder: WanTextEncoder = None):
        self.text_encoder = text_encoder

    ftfy_missing_warned = False

    @staticmethod
    def clean(text):
        """'whitespace' cleaning of PR:9-21, 78: `ftfy.fix_text` (mojibake, curly quotes, full-width forms ... -> canonical
        text), html-unescape twice, collapse spaces.  ftfy is used whenever it is importable; where it is not (this image) the
        first call says so once on stderr — prompts that ftfy would have changed then tokenise differently from the reference
        (plain ASCII prompts, like the 13 examples of the reference, are unaffected: ftfy leaves them as they are)."""
        import html
        import re
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            if not WanPrompter.ftfy_missing_warned:
                import sys
                WanPrompter.ftfy_missing_warned = True
                print("goal_force_amd.WanPrompter: `ftfy` is not installed — prompts are cleaned without ftfy.fix_text "
                      "(wan_prompter.py:11-14); text with mojibake / typographic quotes tokenises differently from the reference",
                      file=sys.stderr)
        text = html.unescape(html.unescape(text)).strip()
        return re.sub(r"\s+", " ", text).strip()

    def tokenize(self, prompt):
        if self.tokenizer is None:
            raise GoalForceError("WanPrompter: no tokenizer loaded (pass tokenizer_path with the google/umt5-xxl files)")
        enc = self.tokenizer([self.clean(prompt)], return_tensors="pt", padding="max_length", truncation=True,
                             max_length=self.text_len, add_special_tokens=True)
        return enc.input_ids, enc.attention_mask

    def encode_ids(self, ids, mask, device="cuda"):
        ids, mask = ids.to(device), mask.to(device)
        seq_lens = mask.gt(0).sum(dim=1).long()
        emb = self.text_encoder(ids, mask)
        for i, v in enumerate(seq_lens):
            emb[:, v:] = 0                                            # PR:107-108 (sic: zeroes every batch row past v)
        return emb

    def encode_prompt(self, prompt, positive=True, device="cuda"):
        ids, mask = self.tokenize(prompt)
        return self.encode_ids(ids, mask, device)
