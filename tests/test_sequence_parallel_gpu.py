"""Head-parallel attention on the GPU: two processes (gloo rendezvous on 127.0.0.1, both on cuda:0 — the GPU box has one
device; RCCL needs one device per rank, gloo carries device tensors) run the tiny model_fn with ControlNet on their token
chunks through the real HIP kernels.  Every op of the path is row-wise except attention, and a head's attention does not
depend on which other heads are in the launch, so the sharded forward must equal the one-process forward BIT FOR BIT.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gen_inputs as gi

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _models():
    from goal_force_amd.controlnet import ControlNet
    from goal_force_amd.dit import WanModel
    cfg = gi.TINY
    dit = WanModel(has_image_input=False, require_clip_embedding=False, **cfg)
    dit.load_state_dict(gi.dit_sd(cfg, seed=41), strict=True)
    cn = ControlNet(gi.TINY_CONTROLNET_LAYERS, dim=cfg["dim"], num_heads=cfg["num_heads"], ffn_dim=cfg["ffn_dim"])
    cn.load_state_dict(gi.controlnet_sd(cfg, gi.TINY_CONTROLNET_LAYERS, seed=42, zero_convs_zero=False), strict=True)
    return dit.to(BF).cuda(), cn.to(BF).cuda()


def _forward(sp, use_flag=False):
    from goal_force_amd.model_fn import ContextCache, model_fn_wan_video
    torch.set_grad_enabled(False)
    dit, cn = _models()
    inp = {k: v.cuda() for k, v in gi.tiny_inputs().items()}
    ts = torch.tensor([995.9], dtype=BF).cuda()
    kw = dict(latents=inp["latents"], timestep=ts, context=inp["ctx_posi"], y=inp["y"], controlnet=cn,
              control_signal_video_latents=inp["control"], elide_zero_controlnet=False)
    if use_flag:
        out = model_fn_wan_video(dit, use_unified_sequence_parallel=True, **kw)      # the reference's own switch
    else:
        out = model_fn_wan_video(dit, sequence_parallel=sp, **kw)
    cache = ContextCache()                                                           # cached cross-attention K/V path
    out_cached = model_fn_wan_video(dit, sequence_parallel=sp, context_cache=cache, **kw)
    return out.cpu(), out_cached.cpu()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from goal_force_amd.sequence_parallel import SequenceParallel
        sp = SequenceParallel()
        a, b = _forward(sp)
        c, _ = _forward(None, use_flag=True)
        torch.save({"out": a, "cached": b, "flag": c}, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_sharded_model_fn_is_bit_identical(tmp_path):
    world = 2                                   # TINY: 72 tokens, 2 heads
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want, want_cached = _forward(None)
    assert torch.equal(want, want_cached)
    # the yardstick is the REFERENCE, not our own unsharded forward: same weights and inputs as golden g5 `model_fn_cn_rand`
    # (the reference's model_fn_wan_video with its ControlNet, fp32 and bf16 runs; tests/golden/make_goldens.py)
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g5_model_fn.npz"))
    f32 = torch.from_numpy(g["model_fn_cn_rand_f32"])
    ref_bf16 = gi.from_u16(g["model_fn_cn_rand_bf16"]).float()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    e_ref = rel(ref_bf16, f32)
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        for k in ("out", "cached", "flag"):
            e = rel(res[k].float(), f32)
            assert e < 1e-2 and e < 2 * e_ref + 1e-4, f"rank {r} {k}: sharded forward vs reference fp32 {e:.3e} (reference bf16 {e_ref:.3e})"
            assert torch.equal(res[k], want), f"rank {r} {k}: sharded forward differs from the one-GPU forward"


def test_world_of_one_is_the_plain_path():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from goal_force_amd.sequence_parallel import SequenceParallel
        sp = SequenceParallel()
        q = torch.randn((128, 256), device="cuda").to(BF)
        from goal_force_amd import ops
        assert torch.equal(sp.attention(q, q, q, 2), ops.flash_attn(q, q, q, 2))
    finally:
        dist.destroy_process_group()


def _vae_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_grad_enabled(False)
        from goal_force_amd.vae import WanVideoVAE
        torch.manual_seed(7)
        vae = WanVideoVAE().to(BF).cuda()
        z = torch.randn((1, 16, 2, 12, 16), generator=torch.Generator().manual_seed(5)).to(BF).cuda()
        got = vae.decode(z, tiled=True, tile_size=(8, 8), tile_stride=(4, 4), tile_group=dist.group.WORLD)
        torch.save(got.cpu(), os.path.join(out, f"v{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_tiled_vae_decode_split_over_a_cfg_pair_is_bit_identical(tmp_path):
    """The decoded tiles of one video split round-robin over the two ranks of a CFG pair (both hold the same latents),
    exchanged, blended by both in the reference's task order: same bits as the one-GPU tiled decode, on both ranks."""
    world = 2
    mp.spawn(_vae_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    torch.set_grad_enabled(False)
    from goal_force_amd.vae import WanVideoVAE
    torch.manual_seed(7)
    vae = WanVideoVAE().to(BF).cuda()
    z = torch.randn((1, 16, 2, 12, 16), generator=torch.Generator().manual_seed(5)).to(BF).cuda()
    want = vae.decode(z, tiled=True, tile_size=(8, 8), tile_stride=(4, 4)).cpu()
    for r in range(world):
        assert torch.equal(torch.load(os.path.join(tmp_path, f"v{r}.pt")), want)
