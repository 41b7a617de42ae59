"""Multi-GPU sharding of the sampling path: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards along two independent axes:
  * samples  — different seeds / CSV rows are independent (what the reference's scripts do with
    `--device_id/--world_size`, scripts/inference/utils.py:25-57): no communication;
  * CFG pair — the cond and uncond forwards of one step are independent given the latents
    (src/goal_force/wan_video_new.py:710-716): ONE exchange of the 4.2 MB noise prediction per step
    inside a 2-rank group, after which both ranks apply the identical CFG + Euler update, so the
    latents stay bit-identical on both without a broadcast.
Layout for world size W >= 2 (even): rank r -> sample r // 2, branch r % 2 (0 = cond, 1 = uncond).
At the end the decoded frames of all samples are all-gathered over the world group.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def split_list_across_devices_contiguous(items: list, world_size: int, device_id: int) -> list:
    """Contiguous CSV sharding of the reference's inference scripts (scripts/inference/utils.py:25-57):
    the first (len % world) ranks get one extra item."""
    n = len(items)
    base, extra = divmod(n, world_size)
    start = device_id * base + min(device_id, extra)
    return items[start:start + base + (1 if device_id < extra else 0)]


class CfgPairParallel:
    """Communication plan for CFG-pair x sample sharding."""

    def __init__(self, rank: int, world_size: int):
        if world_size < 2 or world_size % 2:
            raise ValueError("CFG-pair sharding needs an even world size >= 2")
        self.rank, self.world_size = rank, world_size
        self.sample = rank // 2
        self.branch = rank % 2          # 0 = cond (positive prompt), 1 = uncond
        self.num_samples = world_size // 2
        self.pair_group = None
        # every rank must create every group, in the same order
        for s in range(self.num_samples):
            g = dist.new_group(ranks=[2 * s, 2 * s + 1])
            if s == self.sample:
                self.pair_group = g

    def exchange(self, noise_pred: torch.Tensor):
        """all-gather of this rank's noise prediction inside its CFG pair -> (posi, nega)."""
        buf = [torch.empty_like(noise_pred) for _ in range(2)]
        dist.all_gather(buf, noise_pred.contiguous(), group=self.pair_group)
        return buf[0], buf[1]

    def gather_frames(self, frames: Optional[torch.Tensor], shape, dtype, device) -> List[torch.Tensor]:
        """World all-gather of decoded frames; branch-0 ranks contribute their sample, branch-1 ranks a
        zero tensor that is dropped.  Returns one tensor per sample (on every rank)."""
        mine = frames if (self.branch == 0 and frames is not None) else torch.zeros(shape, dtype=dtype, device=device)
        out = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.world_size)]
        dist.all_gather(out, mine.contiguous())
        return [out[2 * s] for s in range(self.num_samples)]


def init_from_env(backend: Optional[str] = None):
    """torchrun / torch.distributed.run environment -> (rank, local_rank, world_size); initialises the
    default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
