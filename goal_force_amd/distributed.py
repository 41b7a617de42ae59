"""Multi-GPU sharding of the sampling path: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards along two independent axes:
  * samples  — different seeds / CSV rows are independent (what the reference's scripts do with
    `--device_id/--world_size`, scripts/inference/utils.py:25-57): no communication;
  * CFG pair — the cond and uncond forwards of one step are independent given the latents
    (src/goal_force/wan_video_new.py:710-716): ONE exchange of the 4.2 MB noise prediction per step
    inside a 2-rank group, after which both ranks apply the identical CFG + Euler update, so the
    latents stay bit-identical on both without a broadcast.
  * heads     — (optional, degree P) Ulysses attention inside one forward: sequence_parallel.py.
Layout for world size W = samples x 2 x P: rank r -> sp_rank r % P, branch (r // P) % 2 (0 = cond, 1 = uncond),
sample r // 2P; with P = 1: sample r // 2, branch r % 2.
At the end the decoded frames of all samples are all-gathered among the samples' lead ranks (one contributor per sample).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

from ._lib import GoalForceError


def ensure_ipc_env():
    """RCCL hands device buffers between the ranks of one node through HIP IPC handles.  The host driver of the MI355X pool
    only supports the dmabuf form of those handles; with the ROCr default (legacy IPC) `hipIpcGetMemHandle` fails with
    "invalid argument" and the first collective of a multi-rank run dies.  `HSA_ENABLE_IPC_MODE_LEGACY=0` selects dmabuf IPC.
    ROCr reads it when the process first initialises the GPU, so it is set here — `setdefault`: an explicit choice of the
    launcher wins — before anything touches HIP (bench.py, scripts/inference_goal_force.py and init_from_env call this first;
    it never re-executes the process)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def split_list_across_devices_contiguous(items: list, world_size: int, device_id: int) -> list:
    """Contiguous CSV sharding of the reference's inference scripts (scripts/inference/utils.py:25-57):
    the first (len % world) ranks get one extra item."""
    n = len(items)
    base, extra = divmod(n, world_size)
    start = device_id * base + min(device_id, extra)
    return items[start:start + base + (1 if device_id < extra else 0)]


class CfgPairParallel:
    """Communication plan for (samples) x (CFG pair) x (sequence-parallel degree P) sharding.

    rank r -> sp_rank = r % P, branch = (r // P) % 2, sample = r // (2 P).  The P ranks of one (sample, branch) are
    consecutive: they form the sequence-parallel group (head all-to-alls between neighbouring GPUs); the CFG exchange
    runs between the two ranks with equal sp_rank of a sample (each holds the whole noise prediction after the token
    all-gather).  P = 1 is the plain CFG-pair layout: rank r -> sample r // 2, branch r % 2."""

    def __init__(self, rank: int, world_size: int, sp_size: int = 1):
        if sp_size < 1 or world_size < 2 * sp_size or world_size % (2 * sp_size):
            raise ValueError("CFG-pair sharding needs world_size = samples x 2 x sp_size")
        self.rank, self.world_size, self.sp_size = rank, world_size, sp_size
        self.sp_rank = rank % sp_size
        self.branch = (rank // sp_size) % 2          # 0 = cond (positive prompt), 1 = uncond
        self.sample = rank // (2 * sp_size)
        self.num_samples = world_size // (2 * sp_size)
        self.pair_group = None
        self.sp_group = None
        self.sample_group = None      # all 2 P ranks of my sample (frames broadcast)
        self.lead_group = None        # the (branch 0, sp_rank 0) ranks of all samples (frames all-gather); None on the others
        self.is_lead = self.branch == 0 and self.sp_rank == 0
        # every rank must create every group, in the same order
        for s in range(self.num_samples):
            base = 2 * sp_size * s
            for i in range(sp_size):
                g = dist.new_group(ranks=[base + i, base + sp_size + i])
                if s == self.sample and i == self.sp_rank:
                    self.pair_group = g
        if sp_size > 1:
            for s in range(self.num_samples):
                for b in range(2):
                    base = 2 * sp_size * s + sp_size * b
                    g = dist.new_group(ranks=list(range(base, base + sp_size)))
                    if s == self.sample and b == self.branch:
                        self.sp_group = g
            for s in range(self.num_samples):
                g = dist.new_group(ranks=list(range(2 * sp_size * s, 2 * sp_size * (s + 1))))
                if s == self.sample:
                    self.sample_group = g
        else:
            self.sample_group = self.pair_group
        self.lead_ranks = [2 * sp_size * s for s in range(self.num_samples)]
        g = dist.new_group(ranks=self.lead_ranks)
        if self.is_lead:
            self.lead_group = g

    def sequence_parallel(self):
        """The SequenceParallel object of this rank's (sample, branch), or None when P = 1."""
        if self.sp_size == 1:
            return None
        from .sequence_parallel import SequenceParallel
        return SequenceParallel(self.sp_group)

    def exchange(self, noise_pred: torch.Tensor):
        """all-gather of this rank's noise prediction inside its CFG pair -> (posi, nega)."""
        buf = [torch.empty_like(noise_pred) for _ in range(2)]
        dist.all_gather(buf, noise_pred.contiguous(), group=self.pair_group)
        return buf[0], buf[1]

    def gather_frames(self, frames: Optional[torch.Tensor], shape, dtype, device, everywhere: bool = False):
        """End-of-run all-gather of the decoded frames over the LEADS-ONLY group: one contributor per sample (its
        (branch 0, sp_rank 0) rank), nobody sends padding — at 8 GPUs a lead receives 3 x 97 MB instead of every rank receiving
        7 x 97 MB of which half were zeros.  Returns one tensor per sample on the lead ranks and None elsewhere; with
        `everywhere` every lead then broadcasts the list inside its sample's group, so that all ranks return it."""
        out = None
        if self.is_lead:
            if frames is None:
                raise GoalForceError("gather_frames: the lead rank of a sample must pass its frames")
            out = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.num_samples)]
            dist.all_gather(out, frames.contiguous(), group=self.lead_group)
        if everywhere:
            if out is None:
                out = [torch.empty(shape, dtype=dtype, device=device) for _ in range(self.num_samples)]
            for t in out:
                dist.broadcast(t, src=self.lead_ranks[self.sample], group=self.sample_group)
        return out


def init_from_env(backend: Optional[str] = None):
    """torchrun / torch.distributed.run environment -> (rank, local_rank, world_size); initialises the
    default process group when WORLD_SIZE > 1.  Under RCCL ("nccl") every rank needs a GPU of its own: LOCAL_RANK beyond the
    visible devices is an error, not a wrap-around (two ranks on one device would deadlock or silently halve the node).  Only
    GF_DIST_BACKEND=gloo — the single-GPU test transport — lets ranks share a device."""
    ensure_ipc_env()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    have_gpu = torch.cuda.is_available()
    ndev = torch.cuda.device_count() if have_gpu else 0
    if world > 1 and backend is None:
        # GF_DIST_BACKEND=gloo: several ranks sharing one GPU (RCCL wants one device per rank) — used to exercise the
        # N>1 code path on a single-GPU box; gloo carries device tensors through host memory
        backend = os.environ.get("GF_DIST_BACKEND") or ("nccl" if have_gpu else "gloo")
    if world > 1 and backend == "nccl" and local >= ndev:
        raise GoalForceError(f"LOCAL_RANK={local} but only {ndev} GPU(s) are visible: RCCL needs one device per rank "
                             f"(check ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES, or --nproc-per-node)")
    if have_gpu:
        local = local % max(1, ndev)     # gloo on a shared device (tests); a no-op under RCCL after the check above
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
