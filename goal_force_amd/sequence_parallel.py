"""Head-parallel (Ulysses) attention over xGMI: single-video latency on P GPUs.

Behavioural spec: the reference's USP path (diffsynth/distributed/xdit_context_parallel.py:42-131 and
src/goal_force/wan_video_new.py:1526-1585): the token sequence is cut into P contiguous chunks after patchify, every
block runs on its rank's chunk with the RoPE phases of that chunk (xdit:26-40), self-attention exchanges tokens for heads
(xFuserLongContextAttention, Ulysses degree = P), and the head output is all-gathered along the tokens before unpatchify.
The reference wires this for the DiT blocks only — its ControlNet states stay full-length and cannot be added to the
chunked `x` (GF:1489-1522 vs 1565-1570) — so here the ControlNet tokens are sharded the same way ("ControlNet-aware
sharding"): ControlNet block i and DiT block i see the same token chunk and the zero-conv injection stays local.

Per self-attention (q, k, v local [S/P, NH*DH]):
    all-to-all      [S/P, P, (NH/P)*DH] -> [P, S/P, (NH/P)*DH] = all S tokens of this rank's NH/P heads     (x3: q, k, v)
    flash-attention over NH/P heads, S x S                                                               (HIP kernel)
    all-to-all back [P, S/P, (NH/P)*DH] -> [S/P, NH*DH]
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
    def attention(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, num_heads: int) -> torch.Tensor:
        """q, k, v: this rank's tokens [S/P, NH*DH] (already normed and rotated) -> attention output [S/P, NH*DH]."""
        p = self.size
        if p == 1:
            return ops.flash_attn(q, k, v, num_heads)
        if num_heads % p:
            raise GoalForceError(f"sequence parallel: {num_heads} heads do not divide by {p} ranks")
        sl, d = q.shape
        dh = d // p                                   # columns of this rank's head group
        send = torch.empty((3, p, sl, dh), dtype=q.dtype, device=q.device)
        for i, t in enumerate((q, k, v)):             # [S/P, P, dh] -> [P, S/P, dh]: block j goes to rank j
            send[i].copy_(t.reshape(sl, p, dh).transpose(0, 1))
        recv = torch.empty((3, p * sl, dh), dtype=q.dtype, device=q.device)
        works = [dist.all_to_all_single(recv[i], send[i], group=self.group, async_op=True) for i in range(3)]
        for w in works:
            w.wait()
        o_heads = ops.flash_attn(recv[0], recv[1], recv[2], num_heads // p)      # [S, dh]: rows already in token order
        back = torch.empty((p, sl, dh), dtype=q.dtype, device=q.device)
        dist.all_to_all_single(back, o_heads.reshape(p, sl, dh), group=self.group)   # block j = head group j of my tokens
        out = torch.empty((sl, d), dtype=q.dtype, device=q.device)
        out.view(sl, p, dh).copy_(back.transpose(0, 1))
        return out


def attention(sp: Optional[SequenceParallel], q, k, v, num_heads: int):
    """Self-attention entry of the DiT blocks: plain flash-attention without a group, Ulysses with one."""
    if sp is None:
        return ops.flash_attn(q, k, v, num_heads)
    return sp.attention(q, k, v, num_heads)
