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


def preflight(cfgp: CfgPairParallel, device, latent_shape=(1, 16, 21, 60, 104), frames_shape=(81, 480, 832, 3),
              tile_shape=(81, 240, 416, 8), log=None) -> dict:
    """First-contact check of an N > 1 run, BEFORE anything is timed: every collective of the sampling path once at its production
    size — the per-step all-gather of the noise prediction inside the CFG pair (4.19 MB), one VAE-tile broadcast inside the pair
    (129 MB), the end-of-run all-gather of the uint8 frames among the samples' lead ranks (97 MB per sample), and the head
    all-to-all of the sequence-parallel group when P > 1 — each timed, each checked for CONTENT (a rank-dependent fill pattern), so
    that a transport problem (IPC mode, a missing xGMI link, two ranks on one device) surfaces here with the backend's own error
    text instead of as a hang or a wrong number in the timed region.  Also reports the rank -> device map, the backend, and the
    free HBM of every rank.  Returns the report (identical on every rank); raises GoalForceError naming the step that failed."""
    import time
    rep = {"world": cfgp.world_size, "backend": dist.get_backend(), "samples": cfgp.num_samples, "sp_size": cfgp.sp_size, "steps": {}}
    dev = torch.device(device)
    on_gpu = dev.type == "cuda"

    def sync():
        if on_gpu:
            torch.cuda.synchronize(dev)

    def step(name, fn):
        """One collective: run it, then agree on the outcome — every rank all-reduces an ok flag (MIN) after the step, so that a
        content check (or a backend error) on ONE rank raises the same named error on ALL of them, instead of leaving the peers
        in the next step's barrier until the backend's 10-30 minute timeout."""
        err, nbytes, dt = None, 0, 0.0
        try:
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            nbytes = fn()
            sync()
            dt = time.perf_counter() - t0
        except Exception as e:      # noqa: BLE001 — the backend's message is the diagnosis
            err = e
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev if rep["backend"] == "nccl" else "cpu")
        try:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        except Exception:           # noqa: BLE001 — the transport is gone: the local error (if any) is what there is to report
            pass
        where = (f"rank {cfgp.rank} (sample {cfgp.sample}, branch {cfgp.branch}, sp_rank {cfgp.sp_rank}, device {dev}) over backend "
                 f"{rep['backend']}")
        if err is not None:
            if isinstance(err, GoalForceError):
                raise err
            raise GoalForceError(f"multi-GPU pre-flight: `{name}` failed on {where}: {type(err).__name__}: {err}") from err
        if int(ok.item()) == 0:
            raise GoalForceError(f"multi-GPU pre-flight: `{name}` failed on another rank (this is {where}; the failing rank's "
                                 f"message names the cause)")
        rep["steps"][name] = {"bytes": int(nbytes), "seconds": dt}
        if log:
            log(f"  pre-flight rank {cfgp.rank}: {name}: {nbytes / 1e6:.1f} MB in {dt * 1e3:.2f} ms")

    def check(ok, name):
        if not bool(ok):
            raise GoalForceError(f"multi-GPU pre-flight: `{name}` delivered wrong CONTENT on rank {cfgp.rank} (device {dev}, backend "
                                 f"{rep['backend']}): the transport is not moving the peers' data")

    def noise_pair():
        x = torch.full(latent_shape, float(cfgp.branch + 1), dtype=torch.bfloat16, device=dev)
        posi, nega = cfgp.exchange(x)
        check(float(posi.flatten()[0]) == 1.0 and float(nega.flatten()[-1]) == 2.0, "noise-prediction all-gather (CFG pair)")
        return 2 * x.numel() * 2

    def tile_bcast():
        t = torch.full(tile_shape, float(cfgp.sample + 1) if cfgp.branch == 0 else 0.0, dtype=torch.bfloat16, device=dev)
        src = dist.get_global_rank(cfgp.pair_group, 0)
        dist.broadcast(t, src=src, group=cfgp.pair_group)
        check(float(t.flatten()[-1]) == cfgp.sample + 1, "VAE tile broadcast (CFG pair)")
        return t.numel() * 2

    def frames_gather():
        f = torch.full(frames_shape, cfgp.sample + 1, dtype=torch.uint8, device=dev) if cfgp.is_lead else None
        out = cfgp.gather_frames(f, frames_shape, torch.uint8, dev)
        if cfgp.is_lead:
            check(all(int(t.flatten()[-1]) == i + 1 for i, t in enumerate(out)), "frame all-gather (lead ranks)")
            return len(out) * out[0].numel()
        return 0

    # rank -> device map and HBM headroom of every rank FIRST: two ranks on one device under RCCL would fail or hang inside the first
    # collective below, so the duplicate is named here (every rank sees the same list and raises the same error)
    free, total = (torch.cuda.mem_get_info(dev) if on_gpu else (0, 0))
    mine = {"rank": cfgp.rank, "sample": cfgp.sample, "branch": cfgp.branch, "sp_rank": cfgp.sp_rank, "device": str(dev),
            "device_name": torch.cuda.get_device_name(dev) if on_gpu else "cpu", "hbm_free_gb": free / 2 ** 30, "hbm_total_gb": total / 2 ** 30}
    allr = [None] * cfgp.world_size
    dist.all_gather_object(allr, mine)
    if on_gpu and rep["backend"] == "nccl":
        devs = [r["device"] for r in allr]
        if len(set(devs)) != len(devs):
            raise GoalForceError(f"multi-GPU pre-flight: two ranks share a device under RCCL: {devs}")
    step("noise_pred_allgather_pair", noise_pair)
    step("vae_tile_broadcast_pair", tile_bcast)
    step("frames_allgather_leads", frames_gather)
    if cfgp.sp_size > 1:
        sp = cfgp.sequence_parallel()

        def heads_a2a():
            return sp.preflight(dev)
        step("head_all_to_all_sp_group", heads_a2a)
    secs = [None] * cfgp.world_size
    dist.all_gather_object(secs, {k: v["seconds"] for k, v in rep["steps"].items()})
    for r, sc in zip(allr, secs):
        r["seconds"] = sc
    rep["ranks"] = allr
    for name in rep["steps"]:          # the slowest rank's time is the step's time
        rep["steps"][name]["seconds"] = max(r["seconds"][name] for r in allr)
        rep["steps"][name]["bytes"] = max(rep["steps"][name]["bytes"], 0)
    rep["rccl_ranks"] = cfgp.world_size if rep["backend"] == "nccl" else 0
    rep["min_hbm_free_gb"] = min(r["hbm_free_gb"] for r in allr)
    return rep


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
