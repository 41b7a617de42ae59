"""bench.py as the driver launches it (BASELINE config 3's code path): `--gpus 2` = one CFG pair, two fresh
processes with the torch.distributed.run environment, per-step noise-prediction all-gather, tiled VAE decode split over
the pair, end-of-run frame all-gather.  The GPU box has ONE device, so the two ranks share cuda:0 and rendezvous over
gloo (GF_DIST_BACKEND=gloo; RCCL wants one device per rank) — same code path, different transport.
Property checked: sharding the CFG pair over two ranks changes no bit — latents and uint8 frames of the N=2 run equal the
N=1 run's (bench.py prints their sha256 in `self_check`).  World 4 = two videos x CFG pair (BASELINE config 3's layout at half
size): every video's latents and frames — the frames as rank 0 RECEIVED them from the leads-only all-gather — equal the N=1 run
of that video."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
ARGS = ["--layers", "2", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, extra=()):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GF_DIST_BACKEND="gloo")
        if world == 1:
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + ARGS + list(extra),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1500) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][0][-2000:]
    assert not any(l.startswith("{") for so, _ in outs[1:] for l in so.splitlines()), "only rank 0 prints the JSON line"
    return json.loads(lines[0])


def test_bench_two_ranks_bit_identical_to_one():
    one = _launch(1)
    two = _launch(2)
    for j, n in ((one, 1), (two, 2)):
        assert j["n_gpus"] == n and j["steps"] == 2 and j["unit"] == "frames/s" and j["value"] > 0
        assert j["self_check"]["latents_finite"] and j["self_check"]["frames_finite"]
        assert j["roofline"]["achieved"] > 0 and j["roofline_gemm"]
    assert two["distributed"] == {**two["distributed"], "world": 2, "backend": "gloo", "rccl_ranks": 0}
    assert one["distributed"]["world"] == 1 and one["frame_allgather_s"] == 0.0
    assert two["frame_allgather_s"] > 0 and two["samples_gathered"] == 1
    assert two["self_check"]["latents"]["sha256"] == one["self_check"]["latents"]["sha256"]
    assert two["self_check"]["frames_uint8"]["sha256"] == one["self_check"]["frames_uint8"]["sha256"]
    assert "cpu_baseline" not in two


def test_bench_four_ranks_two_videos_bit_identical_to_single_runs():
    four = _launch(4)
    assert four["n_gpus"] == 4 and four["samples_gathered"] == 2 and four["frame_allgather_s"] > 0
    assert four["distributed"]["world"] == 4 and four["distributed"]["backend"] == "gloo"
    per = four["self_check"]["per_sample"]
    assert [d["sample"] for d in per] == [0, 1]
    assert per[0]["frames_uint8_sha256"] != per[1]["frames_uint8_sha256"], "the two videos have different seeds"
    for k in (0, 1):
        one = _launch(1, ["--sample-offset", str(k)])
        (d,) = one["self_check"]["per_sample"]
        assert d["sample"] == k
        assert per[k]["latents_sha256"] == d["latents_sha256"], f"video {k}: latents of the 4-rank run differ from its N=1 run"
        assert per[k]["frames_uint8_sha256"] == d["frames_uint8_sha256"], f"video {k}: gathered frames differ from its N=1 run"
