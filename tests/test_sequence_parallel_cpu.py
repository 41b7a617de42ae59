"""Head-parallel (Ulysses) attention, N>1 logic on CPU: world_size 2 and 4 over gloo.

The exchange (token chunks -> head groups -> token chunks) must reproduce full attention exactly; the HIP attention kernel
is replaced by a per-head fp32 stand-in HERE ONLY (no GPU in this suite) — per-head arithmetic does not depend on which
other heads are in the call, so the sharded result has to be bit-identical to the unsharded one.  Also covered: the
chunk / gather helpers, the RoPE slice, and the (sample, branch, sp_rank) layout with its groups.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

BF = torch.bfloat16
HD = 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _attn_standin(q, k, v, num_heads, scale=None, out=None):
    """softmax(q k^T / sqrt(d)) v per head in fp32 -> bf16 (test-only replacement for ops.flash_attn)."""
    s, d = q.shape
    hd = d // num_heads
    o = torch.empty_like(q)
    for h in range(num_heads):
        sl = slice(h * hd, (h + 1) * hd)
        p = torch.softmax(q[:, sl].float() @ k[:, sl].float().T / hd ** 0.5, dim=-1)
        o[:, sl] = (p @ v[:, sl].float()).to(q.dtype)
    return o


def _qkv(tokens, heads, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn((tokens, heads * HD), generator=g).to(BF) for _ in range(3)]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from goal_force_amd import ops
        from goal_force_amd.dit import RopeTable
        from goal_force_amd.sequence_parallel import SequenceParallel
        ops.flash_attn = _attn_standin                      # test-only stand-in for the HIP kernel (no GPU here)
        sp = SequenceParallel()
        assert (sp.rank, sp.size) == (rank, world)
        heads, tokens = 4, 24
        q, k, v = _qkv(tokens, heads, seed=5)
        ql, kl, vl = (sp.shard_tokens(t).contiguous() for t in (q, k, v))
        assert ql.shape == (tokens // world, heads * HD) and torch.equal(ql, q.chunk(world)[rank])
        o_local = sp.attention(ql, kl, vl, heads)
        o_full = sp.gather_tokens(o_local)
        rope = RopeTable(torch.polar(torch.ones(tokens, 1, 4, dtype=torch.float64),
                                     torch.arange(tokens * 4, dtype=torch.float64).view(tokens, 1, 4)), "cpu")
        rl = sp.shard_rope(rope)
        sl = tokens // world
        assert rl.tokens == sl and torch.equal(rl.cos, rope.cos[rank * sl:(rank + 1) * sl])
        assert sp.local_tokens(tokens + 1) == -(-(tokens + 1) // world)      # ragged splits: torch.chunk's ceil(S / P)
        torch.save({"o_local": o_local, "o_full": o_full}, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ulysses_attention_equals_full_attention(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    q, k, v = _qkv(24, 4, seed=5)
    want = _attn_standin(q, k, v, 4)
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert torch.equal(res["o_full"], want), "gathered head-parallel attention must equal full attention bit for bit"
        assert torch.equal(res["o_local"], want.chunk(world)[r])


def _layout_worker(rank, world, port, sp_size, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from goal_force_amd.distributed import CfgPairParallel
        cp = CfgPairParallel(rank, world, sp_size=sp_size)
        assert cp.sp_rank == rank % sp_size and cp.branch == (rank // sp_size) % 2 and cp.sample == rank // (2 * sp_size)
        sp = cp.sequence_parallel()
        assert sp is not None and (sp.rank, sp.size) == (cp.sp_rank, sp_size)
        # the sp group is exactly the ranks of my (sample, branch): gather the global ranks over it
        mine = torch.tensor([float(rank)])
        got = sp.gather_tokens(mine)
        base = 2 * sp_size * cp.sample + sp_size * cp.branch
        assert got.tolist() == [float(base + i) for i in range(sp_size)]
        # the CFG exchange pairs equal sp_ranks of the two branches
        posi, nega = cp.exchange(torch.full((2,), float(rank)).to(BF))
        b0 = 2 * sp_size * cp.sample + cp.sp_rank
        assert float(posi[0]) == b0 and float(nega[0]) == b0 + sp_size
        frames = cp.gather_frames(torch.full((3,), float(cp.sample + 1)) if (cp.branch == 0 and cp.sp_rank == 0) else None,
                                  (3,), torch.float32, "cpu", everywhere=True)
        assert [float(f[0]) for f in frames] == [float(s + 1) for s in range(cp.num_samples)]
        open(os.path.join(out, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_layout_samples_x_cfg_pair_x_sp(tmp_path):
    world, sp_size = 4, 2
    mp.spawn(_layout_worker, args=(world, _free_port(), sp_size, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, f"ok{r}")) for r in range(world))


def test_layout_rejects_bad_world():
    from goal_force_amd.distributed import CfgPairParallel
    with pytest.raises(ValueError):
        CfgPairParallel(0, 6, sp_size=2)
    with pytest.raises(ValueError):
        CfgPairParallel(0, 2, sp_size=2)


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from goal_force_amd.training import allreduce_gradients
        g = torch.Generator().manual_seed(11)
        params = [torch.nn.Parameter(torch.randn(shape, generator=g).to(BF)) for shape in ((7, 5), (33,), (4, 4, 2))]
        frozen = torch.nn.Parameter(torch.ones(3).to(BF), requires_grad=False)
        for i, p in enumerate(params):
            p.grad = torch.full(p.shape, float(rank + 1 + i)).to(BF)
        allreduce_gradients(params + [frozen], bucket_bytes=64)          # several buckets
        for i, p in enumerate(params):
            want = sum(float(r + 1 + i) for r in range(world)) / world
            assert torch.equal(p.grad, torch.full(p.shape, want).to(BF)), (i, p.grad.flatten()[:3])
        assert frozen.grad is None
        open(os.path.join(out, f"ddp{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_average(tmp_path):
    world = 2
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, f"ddp{r}")) for r in range(world))


def _ragged_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from goal_force_amd import ops
        from goal_force_amd.dit import RopeTable
        from goal_force_amd.sequence_parallel import SequenceParallel
        ops.flash_attn = _attn_standin
        sp = SequenceParallel()
        heads, tokens = 4, 23                                   # 23 tokens over 4 ranks: chunks of 6, the last one 5 + 1 pad row
        sl = sp.local_tokens(tokens)
        assert sl == 6
        q, k, v = _qkv(tokens, heads, seed=9)
        ql, kl, vl = (sp.shard_tokens(t).contiguous() for t in (q, k, v))
        assert ql.shape[0] == sl
        if rank == world - 1:
            assert float(ql[5].abs().sum()) == 0 and torch.equal(ql[:5], q[18:])      # zero pad row (xdit:76-79)
        rope = RopeTable(torch.polar(torch.ones(tokens, 1, 4, dtype=torch.float64),
                                     torch.arange(tokens * 4, dtype=torch.float64).view(tokens, 1, 4)), "cpu")
        rl = sp.shard_rope(rope)
        if rank == world - 1:
            assert torch.equal(rl.cos[5], torch.ones(4)) and torch.equal(rl.sin[5], torch.zeros(4))   # pad_freqs: ones
        o = sp.gather_tokens(sp.attention(ql, kl, vl, heads), total=tokens)
        assert o.shape[0] == tokens
        torch.save(o, os.path.join(out, f"r{rank}.pt"))
        with pytest.raises(Exception):
            sp.local_tokens(9)                                  # 9 tokens / 4 ranks: chunks of 3 leave rank 3 empty (torch.chunk gives 3 chunks)
    finally:
        dist.destroy_process_group()


def test_ulysses_ragged_token_count_follows_the_reference_padding(tmp_path):
    """S % P != 0 (xdit_context_parallel.py:15-40, 75-79, 103): chunks of ceil(S/P), zero pad rows, unit RoPE phases for them, the
    pad rows take part in the attention as keys, and the gathered output is cut back to S — i.e. the sharded attention equals
    FULL attention over the zero-padded sequence, rows [:S]."""
    world = 4
    mp.spawn(_ragged_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    q, k, v = _qkv(23, 4, seed=9)
    pad = lambda t: torch.cat([t, torch.zeros((1, t.shape[1]), dtype=t.dtype)])
    want = _attn_standin(pad(q), pad(k), pad(v), 4)[:23]
    for r in range(world):
        assert torch.equal(torch.load(os.path.join(tmp_path, f"r{r}.pt")), want)
    assert not torch.equal(want, _attn_standin(q, k, v, 4)), "the pad row IS a key: the result differs from attention over 23 tokens"
