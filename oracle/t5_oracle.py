"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Functional torch-CPU restatement of the umT5 encoder forward
(diffsynth/models/wan_video_text_encoder.py: T5LayerNorm 22-35, T5Attention 38-93, T5FeedForward 96-111,
T5SelfAttention 114-144, T5RelativeEmbedding 147-190, WanTextEncoder.forward 245-255) and of the prompter's
zeroing (diffsynth/prompters/wan_prompter.py:99-109).  Pinned by tests/golden/g8_text_encoder.npz."""
import math

import torch
import torch.nn.functional as F


def t5_norm(x, w, eps=1e-6):
    x = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        x = x.type_as(w)
    return w * x


def gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def rel_bias(emb, lq, lk, num_buckets=32, max_dist=128):
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    me = nb // 2
    large = me + (torch.log(rel.float() / me) / math.log(max_dist / me) * (nb - me)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    out = out + torch.where(rel < me, rel, large)
    return emb[out].permute(2, 0, 1).unsqueeze(0).contiguous()


def encode(ids, mask, sd, num_heads, num_layers):
    x = sd["token_embedding.weight"][ids]
    b = x.shape[0]
    for i in range(num_layers):
        p = f"blocks.{i}."
        e = rel_bias(sd[p + "pos_embedding.embedding.weight"], x.shape[1], x.shape[1])
        h = t5_norm(x, sd[p + "norm1.weight"])
        c = sd[p + "attn.q.weight"].shape[0] // num_heads
        q = F.linear(h, sd[p + "attn.q.weight"]).view(b, -1, num_heads, c)
        k = F.linear(h, sd[p + "attn.k.weight"]).view(b, -1, num_heads, c)
        v = F.linear(h, sd[p + "attn.v.weight"]).view(b, -1, num_heads, c)
        bias = x.new_zeros(b, num_heads, q.size(1), k.size(1)) + e
        if mask is not None:
            bias.masked_fill_(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
        attn = torch.einsum("binc,bjnc->bnij", q, k) + bias
        attn = F.softmax(attn.float(), dim=-1).type_as(attn)
        a = torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, -1, num_heads * c)
        x = x + F.linear(a, sd[p + "attn.o.weight"])
        h = t5_norm(x, sd[p + "norm2.weight"])
        f = F.linear(h, sd[p + "ffn.fc1.weight"]) * gelu(F.linear(h, sd[p + "ffn.gate.0.weight"]))
        x = x + F.linear(f, sd[p + "ffn.fc2.weight"])
    return t5_norm(x, sd["norm.weight"])


def encode_prompt_ids(ids, mask, sd, num_heads, num_layers):
    emb = encode(ids, mask, sd, num_heads, num_layers)
    for i, v in enumerate(mask.gt(0).sum(dim=1).long()):
        emb[:, v:] = 0
    return emb
