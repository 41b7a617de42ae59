"""Head-parallel (Ulysses) attention over xGMI: single-video latency on P GPUs.

Behavioural spec: the reference's USP path (diffsynth/distributed/xdit_context_parallel.py:42-131 and
src/goal_force/wan_video_new.py:1526-1585): the token sequence is cut into P contiguous chunks after patchify, every
block runs on its rank's chunk with the RoPE phases of that chunk (xdit:26-40), self-attention exchanges tokens for heads
(xFuserLongContextAttention, Ulysses degree = P), and the head output is all-gathered along the tokens before unpatchify.
The reference wires this for the DiT blocks only — its ControlNet states stay full-length and cannot be added to the
chunked `x` (GF:1489-1522 vs 1565-1570) — so here the ControlNet tokens are sharded the same way ("ControlNet-aware
sharding"): ControlNet block i and DiT block i see the same token chunk and the zero-conv injection stays local.

Per self-attention (q, k, v local [S/P, NH*DH]):
    all-to-all      [S/P, P, (NH/P)*DH] -> [P, S/P, (NH/P)*DH] = all S tokens of this rank's NH/P heads     (x3: k, v, q)
    flash-attention over NH/P heads, S x S, in two halves of the head group                              (HIP kernel)
    all-to-all back [P, S/P, (NH/P)*DH] -> [S/P, NH*DH]                                                    (x2: one per half)
The exchanges are asynchronous and ordered so that they hide under compute (SelfAttention.attend): K's exchange starts as soon as K
is normed and rotated and flies under the V projection, V's under the Q projection and its norm; the first half's way back flies
under the second half's attention.  Exposed per attention: Q's exchange and the second half's way back (2 of 5 transfers).
The V-by-GEMM shortcut of the one-GPU path (gf_linear_vt32: the V projection writes the attention kernel's V^T operand) does not
carry over: the projection mixes all channels of a rank's OWN tokens, the attention needs ALL tokens of the rank's heads, so V must
cross the fabric between the two — the received V goes through gf_transpose_v32 (0.7 % of the attention's time).
Full-width RMSNorm and RoPE act per token and run before the exchange on the local chunk; cross-attention keeps all heads
local (its K/V are the 512 replicated text tokens): no communication.  Every head is computed by the same kernel on the same
values in the same key order as on one GPU and every other op is row-wise, so the sharded forward is bit-identical to the
single-GPU forward.

Traffic per attention and rank: 4 x (S/P) x D x 2 B x (P-1)/P  (= 147 MB at S=32760, D=5120, P=8), point to point over
xGMI: RCCL's all-to-all keeps all P-1 links of a GPU busy at once, which is the pattern xGMI is good at (no ring).
Token counts that do not divide by P follow the reference (xdit_context_parallel.py:15-40, 75-79, 103): `torch.chunk` cuts
chunks of ceil(S / P) tokens, the short last chunk is padded with ZERO rows, their RoPE phases are padded with ones (no
rotation), the pad rows run through every block like tokens — in the self-attention they are keys like any other — and are cut
off after the final token all-gather.  A sharded forward then equals the one-GPU forward over the padded sequence (tested),
not the one-GPU forward over S tokens; 32760 divides by 2, 4 and 8, so BASELINE config 3 never pads.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from ._lib import GoalForceError


class SequenceParallel:
    """One sequence-parallel group: `group` is a torch.distributed group of P ranks (None = the world group)."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise GoalForceError("SequenceParallel needs an initialised torch.distributed process group")
        self.group = group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    # ---- token sharding -------------------------------------------------------------------------------------------
    def local_tokens(self, total: int) -> int:
        """Tokens per rank = torch.chunk's chunk size ceil(total / P) (xdit_context_parallel.py:75).  Like the reference, a split
        that would leave a rank without any real token is refused (torch.chunk returns fewer than P chunks there and the
        reference's `chunks[rank]` raises)."""
        sl = -(-total // self.size)
        if sl * (self.size - 1) >= total and self.size > 1:
            raise GoalForceError(f"sequence parallel: {total} tokens leave rank {self.size - 1} of {self.size} without a token")
        return sl

    def shard_tokens(self, x2: torch.Tensor) -> torch.Tensor:
        """[S, D] -> this rank's contiguous chunk [ceil(S/P), D] (torch.chunk order as in the reference, GF:1528-1531); a short
        last chunk is padded with zero rows (xdit_context_parallel.py:76-79).  A view when nothing has to be padded."""
        sl = self.local_tokens(x2.shape[0])
        part = x2[self.rank * sl:(self.rank + 1) * sl]
        if part.shape[0] == sl:
            return part
        out = torch.zeros((sl,) + tuple(x2.shape[1:]), dtype=x2.dtype, device=x2.device)
        out[:part.shape[0]] = part
        return out

    def shard_rope(self, rope):
        """RoPE phases of this rank's tokens (xdit_context_parallel.py:36-37); pad rows get the unit phase (pad_freqs pads the
        complex table with ones, xdit:15-25): cos 1, sin 0."""
        from .dit import RopeTable
        sl = self.local_tokens(rope.tokens)
        out = RopeTable.__new__(RopeTable)
        cos, sin = rope.cos[self.rank * sl:(self.rank + 1) * sl], rope.sin[self.rank * sl:(self.rank + 1) * sl]
        if cos.shape[0] < sl:
            pad = sl - cos.shape[0]
            cos = torch.cat([cos, torch.ones((pad, cos.shape[1]), dtype=cos.dtype, device=cos.device)])
            sin = torch.cat([sin, torch.zeros((pad, sin.shape[1]), dtype=sin.dtype, device=sin.device)])
        out.cos, out.sin = cos.contiguous(), sin.contiguous()
        out.tokens = sl
        return out

    def gather_tokens(self, x_local: torch.Tensor, total: Optional[int] = None) -> torch.Tensor:
        """[ceil(S/P), C] per rank -> [S, C] on every rank (get_sp_group().all_gather(x, dim=1), GF:1584); `total` = S cuts the
        pad rows off again (xdit_context_parallel.py:103)."""
        x_local = x_local.contiguous()
        out = torch.empty((x_local.shape[0] * self.size,) + tuple(x_local.shape[1:]), dtype=x_local.dtype,
                          device=x_local.device)
        dist.all_gather_into_tensor(out, x_local, group=self.group)
        return out if total is None or total == out.shape[0] else out[:total]

    # ---- attention ------------------------------------------------------------------------------------------------
    def heads_start(self, t: torch.Tensor, num_heads: Optional[int] = None):
        """Start the tokens-for-heads exchange of ONE projection: t = this rank's tokens [S/P, NH*DH] -> (work, recv) with
        recv [S, (NH/P)*DH] = all S tokens of this rank's head group once `work.wait()` has returned.  The collective is
        asynchronous (RCCL runs it on the process group's own stream, ordered after the kernels already queued on the current one):
        SelfAttention.attend starts K's exchange, then computes the V projection while it flies, and so on — only the last
        exchange (Q) is exposed.  `num_heads`: checked against the group size HERE, before any fabric traffic is issued."""
        p = self.size
        sl, d = t.shape
        if d % p or (num_heads is not None and (num_heads % p or d % num_heads)):
            raise GoalForceError(f"sequence parallel: {num_heads if num_heads is not None else '?'} heads of total width {d} do not "
                                 f"divide over {p} ranks")
        dh = d // p
        send = torch.empty((p, sl, dh), dtype=t.dtype, device=t.device)
        send.copy_(t.reshape(sl, p, dh).transpose(0, 1))      # [S/P, P, dh] -> [P, S/P, dh]: block j goes to rank j
        recv = torch.empty((p * sl, dh), dtype=t.dtype, device=t.device)
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        return work, recv, send                                # (send is kept alive until the wait)

    def attention_started(self, hq, hk, hv, num_heads: int, out_shape, scale=None) -> torch.Tensor:
        """The attention over this rank's head group from three started exchanges (heads_start), and the heads-for-tokens exchange
        back.  The head group is computed in two halves: the first half's way back flies under the second half's attention."""
        p = self.size
        if num_heads % p:
            raise GoalForceError(f"sequence parallel: {num_heads} heads do not divide by {p} ranks")
        sl, d = out_shape
        hl = num_heads // p                            # heads of this rank
        hd = d // num_heads
        for h in (hk, hv, hq):
            h[0].wait()
        q, k, v = hq[1], hk[1], hv[1]                  # [S, hl * hd], rows in token order
        parts = [(0, hl)] if hl < 2 else [(0, (hl + 1) // 2), ((hl + 1) // 2, hl)]
        backs = []
        for h0, h1 in parts:
            c0, c1 = h0 * hd, h1 * hd
            o = ops.flash_attn(q[:, c0:c1], k[:, c0:c1], v[:, c0:c1], h1 - h0, scale=scale)   # [S, (h1 - h0) * hd]
            back = torch.empty((p, sl, c1 - c0), dtype=o.dtype, device=o.device)
            backs.append((dist.all_to_all_single(back, o.reshape(p, sl, c1 - c0), group=self.group, async_op=True), back, o, c0, c1))
        out = torch.empty((sl, d), dtype=q.dtype, device=q.device)
        ov = out.view(sl, p, hl * hd)                  # column j * (hl hd) + c = head group j (from rank j), its column c
        for w, back, _, c0, c1 in backs:
            w.wait()
            ov[:, :, c0:c1].copy_(back.transpose(0, 1))
        return out

    def attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int, scale=None) -> torch.Tensor:
        """q, k, v: this rank's tokens [S/P, NH*DH] (already normed and rotated) -> attention output [S/P, NH*DH]."""
        if self.size == 1:
            return ops.flash_attn(q, k, v, num_heads, scale=scale)
        if num_heads % self.size:
            raise GoalForceError(f"sequence parallel: {num_heads} heads do not divide by {self.size} ranks")
        hk, hv, hq = self.heads_start(k, num_heads), self.heads_start(v, num_heads), self.heads_start(q, num_heads)
        return self.attention_started(hq, hk, hv, num_heads, tuple(q.shape), scale=scale)

    def preflight(self, device) -> int:
        """One head all-to-all at production size (S = 32760 tokens, D = 5120: [S/P, D] bf16 per rank) with a content check;
        returns the bytes this rank sent (distributed.preflight)."""
        p = self.size
        sl, d = -(-32760 // p), 5120
        t = torch.full((sl, d), float(self.rank + 1), dtype=torch.bfloat16, device=device)
        w, recv, _ = self.heads_start(t)
        w.wait()
        want = torch.arange(1, p + 1, dtype=torch.float32, device=recv.device).repeat_interleave(sl)
        if not torch.equal(recv[:, 0].float(), want):
            raise GoalForceError("sequence-parallel pre-flight: the head all-to-all delivered the wrong blocks")
        return t.numel() * 2 * (p - 1) // p


def attention(sp: Optional[SequenceParallel], q, k, v, num_heads: int):
    """Self-attention entry of the DiT blocks: plain flash-attention without a group, Ulysses with one."""
    if sp is None:
        return ops.flash_attn(q, k, v, num_heads)
    return sp.attention(q, k, v, num_heads)
