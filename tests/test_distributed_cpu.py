"""N>1 path on CPU: world_size 2 and 4 over gloo.  Checks the CFG-pair exchange, that both ranks of a pair
end with bit-identical latents equal to the sequential (1-process) loop, the frame all-gather, and the
reference's contiguous CSV sharding."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goal_force_amd.distributed import CfgPairParallel, split_list_across_devices_contiguous

BF = torch.bfloat16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_model_fn(dit=None, controlnet=None, latents=None, timestep=None, context=None, **kw):
    # deterministic stand-in for the denoiser: depends on latents, prompt and timestep
    return (latents.float() * 0.5 + context.float().mean() + timestep.float() * 1e-3).to(BF)


def _cpu_cfg_euler(latents, posi, nega, cfg, dsigma):
    pred = posi if nega is None else nega + cfg * (posi - nega)
    latents.copy_(latents + pred * torch.tensor(dsigma))
    return latents


def _run_loop(cfg_parallel, seed):
    from goal_force_amd import pipeline as pl
    pl.ops.cfg_euler_step = _cpu_cfg_euler          # test-only stand-in for the HIP kernel (no GPU here)
    pipe = pl.WanVideoPipeline(device="cpu")
    pipe.dit = object()
    pipe.model_fn = _fake_model_fn
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn((1, 16, 3, 4, 6), generator=g).to(BF)
    cp = torch.randn((1, 7, 8), generator=g).to(BF)
    cn = torch.randn((1, 7, 8), generator=g).to(BF)
    return pipe.denoise(lat, cp, cn, None, None, num_inference_steps=4, cfg_scale=5.0, controlnet=False,
                        cfg_parallel=cfg_parallel)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cp = CfgPairParallel(rank, world)
        assert (cp.sample, cp.branch, cp.num_samples) == (rank // 2, rank % 2, world // 2)
        mine = torch.full((2, 3), float(rank)).to(BF)
        posi, nega = cp.exchange(mine)
        assert float(posi[0, 0]) == 2 * cp.sample and float(nega[0, 0]) == 2 * cp.sample + 1
        lat = _run_loop(cp, seed=100 + cp.sample)
        lead_only = cp.gather_frames(lat if cp.branch == 0 else None, lat.shape, lat.dtype, "cpu")
        assert (lead_only is None) == (cp.branch == 1), "only the samples' lead ranks take part in the frame all-gather"
        frames = cp.gather_frames(lat if cp.branch == 0 else None, lat.shape, lat.dtype, "cpu", everywhere=True)
        assert len(frames) == world // 2
        if lead_only is not None:
            assert all(torch.equal(a, b) for a, b in zip(lead_only, frames))
        torch.save({"lat": lat, "frames": frames}, os.path.join(out, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cfg_pair_parallel_matches_sequential(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    for s in range(world // 2):
        seq = _run_loop(None, seed=100 + s)
        a, b = res[2 * s]["lat"], res[2 * s + 1]["lat"]
        assert torch.equal(a, b), "the two ranks of a CFG pair must hold bit-identical latents"
        assert torch.equal(a, seq), "pair-parallel loop must equal the sequential loop bit for bit"
        for r in range(world):
            assert torch.equal(res[r]["frames"][s], seq)


def test_contiguous_csv_sharding_matches_reference_rule():
    items = list("abcde")
    assert split_list_across_devices_contiguous(items, 2, 0) == ["a", "b", "c"]   # docstring example of the reference
    assert split_list_across_devices_contiguous(items, 2, 1) == ["d", "e"]
    for n in range(0, 14):
        for w in range(1, 9):
            parts = [split_list_across_devices_contiguous(list(range(n)), w, d) for d in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_cfg_pair_needs_even_world():
    with pytest.raises(ValueError):
        CfgPairParallel(0, 3)


def _preflight_worker(rank, world, port, out, sp_size, sabotage):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from goal_force_amd import distributed as gd
        cp = CfgPairParallel(rank, world, sp_size=sp_size)
        if sabotage and rank == 1:
            # a transport that "succeeds" without moving the peer's data (the collective is entered, what it delivers is this rank's
            # own buffer twice): the content check must catch it
            real = cp.exchange
            cp.exchange = lambda x: (real(x), (x, x))[1]
        try:
            rep = gd.preflight(cp, "cpu", latent_shape=(1, 16, 3, 4, 6), frames_shape=(5, 8, 8, 3), tile_shape=(5, 8, 8, 8))
            torch.save({"rep": rep, "error": None}, os.path.join(out, f"r{rank}.pt"))
        except Exception as e:      # noqa: BLE001
            torch.save({"rep": None, "error": f"{type(e).__name__}: {e}"}, os.path.join(out, f"r{rank}.pt"))
            if not sabotage:
                raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sp_size", [(2, 1), (4, 1), (8, 1), (4, 2)])
def test_multi_gpu_preflight_runs_every_collective_and_reports_the_layout(world, sp_size, tmp_path):
    """bench.py --gpus N runs distributed.preflight before its timed region: every collective of the path once,
    content-checked, with the rank -> device map and per-rank memory head-room in the report (identical on all ranks)."""
    mp.spawn(_preflight_worker, args=(world, _free_port(), str(tmp_path), sp_size, False), nprocs=world, join=True)
    reps = [torch.load(os.path.join(tmp_path, f"r{r}.pt"))["rep"] for r in range(world)]
    want = {"noise_pred_allgather_pair", "vae_tile_broadcast_pair", "frames_allgather_leads"} | ({"head_all_to_all_sp_group"} if sp_size > 1 else set())
    for r, rep in enumerate(reps):
        assert set(rep["steps"]) == want and rep["world"] == world and rep["backend"] == "gloo" and rep["rccl_ranks"] == 0
        assert rep["samples"] == world // (2 * sp_size)
        assert [x["rank"] for x in rep["ranks"]] == list(range(world))
        assert [(x["sample"], x["branch"], x["sp_rank"]) for x in rep["ranks"]] == \
            [(q // (2 * sp_size), (q // sp_size) % 2, q % sp_size) for q in range(world)]
        assert rep["steps"]["noise_pred_allgather_pair"]["bytes"] == 2 * 16 * 3 * 4 * 6 * 2
        assert all(v["seconds"] >= 0 for v in rep["steps"].values())


def test_multi_gpu_preflight_fails_loudly_on_a_transport_that_moves_nothing(tmp_path):
    mp.spawn(_preflight_worker, args=(2, _free_port(), str(tmp_path), 1, True), nprocs=2, join=True)
    err = torch.load(os.path.join(tmp_path, "r1.pt"))["error"]
    assert err and "pre-flight" in err and "noise-prediction all-gather" in err and "rank 1" in err, err
    # ... and the peer raises too, at the same step, instead of waiting in the next barrier for the backend's timeout
    err0 = torch.load(os.path.join(tmp_path, "r0.pt"))["error"]
    assert err0 and "`noise_pred_allgather_pair` failed on another rank" in err0 and "this is rank 0" in err0, err0
