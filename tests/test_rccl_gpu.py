"""RCCL on the GPU box.  The box has ONE device, and RCCL wants one device per rank, so the multi-rank paths are tested
over gloo (test_bench_gpu.py, test_sequence_parallel_gpu.py, test_distributed_cpu.py).  What CAN be checked here is that the
RCCL backend itself initialises in this environment and accepts every collective the package issues, with the argument
forms and tensor kinds it issues them with (a one-rank world: each collective degenerates to a copy):

  distributed.CfgPairParallel.exchange       all_gather (list form) of the bf16 noise prediction in a new_group
  distributed.CfgPairParallel.gather_frames  all_gather of uint8 frames in the leads-only group + broadcast in the sample group
  sequence_parallel.SequenceParallel         all_to_all_single (async) + all_gather_into_tensor
  vae tiled decode over a CFG pair           broadcast
  bench.py                                   barrier, all_reduce(MAX)

A child process does it (the default process group is per process, and a failing RCCL init must not take pytest down)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["GF_ROOT"])
import torch, torch.distributed as dist
from goal_force_amd.distributed import init_from_env
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
g = dist.new_group(ranks=[0])
BF = torch.bfloat16
# CFG exchange: list all_gather of the noise prediction [1,16,21,60,104] bf16 inside a sub-group
n = torch.randn((1, 16, 21, 60, 104), device="cuda").to(BF)
buf = [torch.empty_like(n)]
dist.all_gather(buf, n.contiguous(), group=g)
assert torch.equal(buf[0], n)
# frames: uint8 [81, 480, 832, 3] over the world group
f = torch.randint(0, 256, (81, 480, 832, 3), dtype=torch.uint8, device="cuda")
out = [torch.empty_like(f)]
dist.all_gather(out, f)
assert torch.equal(out[0], f)
# head-parallel attention: async all_to_all_single + all_gather_into_tensor through the package's own class
from goal_force_amd.sequence_parallel import SequenceParallel
from goal_force_amd import ops
sp = SequenceParallel(g)
q = torch.randn((256, 256), device="cuda").to(BF)
assert torch.equal(sp.attention(q, q, q, 2), ops.flash_attn(q, q, q, 2))
assert torch.equal(sp.gather_tokens(sp.shard_tokens(q)), q)
send, recv = q.reshape(1, 256, 256).contiguous(), torch.empty((1, 256, 256), dtype=BF, device="cuda")
w = dist.all_to_all_single(recv, send, group=g, async_op=True)
w.wait()
assert torch.equal(recv, send)
# VAE tile exchange, bench bookkeeping
t = torch.randn((3, 8, 16, 16), device="cuda").to(BF)
dist.broadcast(t, src=0, group=g)
dist.barrier()
m = torch.tensor([1.25], device="cuda")
dist.all_reduce(m, op=dist.ReduceOp.MAX)
assert float(m) == 1.25
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_backend_accepts_every_collective_of_the_package():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # HSA_ENABLE_IPC_MODE_LEGACY=0: dmabuf IPC, the only form this pool's host driver supports (distributed.ensure_ipc_env sets
    # it by default; spelled out here because the child initialises the GPU before it imports the package)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GF_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, f"stdout: {r.stdout[-2000:]}\nstderr: {r.stderr[-4000:]}"
