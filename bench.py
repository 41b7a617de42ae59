#!/usr/bin/env python3
"""bench.py — Goal-Force denoising benchmark on MI355X (BASELINE.json: denoise-step ms + frames/sec,
Wan2.2-I2V-A14B 832x480x81f, 50 steps).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work for N > 1.  Started WITHOUT a launcher (no WORLD_SIZE in the environment) `--gpus N` starts its own N ranks:
the parent — before it imports torch or touches a GPU — spawns N fresh child processes of this file with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, relays rank 0's single JSON line on its stdout
(the other ranks' stdout goes to stderr) and exits non-zero if any child fails (the reference's scripts shard the same way,
one plain process per device: scripts/inference/utils.py:25-57).

A "step" is one full denoising iteration of the reference loop (src/goal_force/wan_video_new.py:697-723):
cond + uncond model_fn forwards (40 DiT blocks each, + 10 ControlNet blocks and 10 zero-conv GEMMs on
high-noise steps), CFG combine and the flow-match Euler update, at the full 832x480x81f size
(latents [1,16,21,60,104], 32760 video tokens, 512 text tokens), both A14B-sized experts and both
ControlNets resident, random-init bf16 weights, synthetic inputs already in HBM.

The K timed steps are spread evenly over the real 50-step / shift-5 schedule so they contain the
schedule's mix of high-noise (with ControlNet) and low-noise (zero ControlNet2 elided, bit-identical)
steps.  value = frames/s = videos * 81 / (50 * seconds_per_step + measured tiled VAE decode seconds).

N = 1: one video, CFG pair evaluated sequentially.  N >= 2 (even): rank r -> video r//2, CFG branch r%2,
one RCCL all-gather of the 4.2 MB noise prediction per step inside each pair (weak scaling in videos).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL's intra-node transport needs dmabuf IPC handles on this driver (goal_force_amd/distributed.py::ensure_ipc_env); ROCr reads
# the variable when the process first touches the GPU, so it is set before torch is imported.  setdefault: the launcher's choice wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# HBM-side bytes per launch of the dominant kernels come from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes over
# tools/microbench.py; (2 x FETCH_SIZE [gfx950 reports half of a 16-B/lane stream] + WRITE_SIZE) KiB -> bytes).  PMC needs the
# profiler, so the bench line carries STATIC figures ("traffic_static": true) read from profiles/pmc_static.json — which records
# the sha256 of the kernel sources the passes were taken on (tools/pmc_static.py).  A figure whose kernel source has changed since
# is NOT printed: traffic becomes null and the line says why, loudly, instead of describing a kernel that no longer exists.
PMC_STATIC = os.path.join(ROOT, "profiles", "pmc_static.json")


def source_sha256(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "goal_force_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def static_traffic(key):
    """(bytes or None, source note) for `key` of profiles/pmc_static.json ("attn_self", "gemm_ffn1", "gemm_ffn2", "gemm_dd",
    "gemm_fp8_*"): None with the reason when the file is absent, the entry missing, or the kernel sources differ from the ones
    the counters were collected on."""
    try:
        with open(PMC_STATIC) as f:
            tab = json.load(f)
        e = tab["entries"][key]
    except (OSError, KeyError, ValueError) as ex:
        return None, f"no static PMC figure for {key} ({type(ex).__name__})"
    now = source_sha256(e["sources"])
    if now != e["sources_sha256"]:
        msg = (f"STALE: {', '.join(e['sources'])} changed since the PMC pass {e['from']} (sha256 {e['sources_sha256'][:12]} -> "
               f"{now[:12]}): re-run tools/profile_r05.sh + tools/pmc_static.py")
        print(f"bench.py: roofline traffic for {key} withheld — {msg}", file=sys.stderr)
        return None, msg
    return int(e["bytes_per_launch"]), f"{e['from']} (kernel sources sha256 {now[:12]}, library {tab.get('gf_version', '?')})"


PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (Chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA peak (same table): the denominator of the GEMM entries under --fp8 (config 5)
S_TOK, DIM, HEADS, FFN, LTXT = 32760, 5120, 40, 13824, 512


def block_flops(s, l=LTXT, d=DIM, f=FFN):
    return 12 * s * d * d + 4 * l * d * d + 4 * s * d * f + 4 * s * s * d + 4 * s * l * d


def forward_flops(with_controlnet):
    fl = 40 * block_flops(S_TOK) + 2 * S_TOK * 144 * DIM + 2 * S_TOK * DIM * 64
    if with_controlnet:
        fl += 10 * block_flops(S_TOK) + 10 * 2 * S_TOK * DIM * DIM + 2 * S_TOK * 64 * DIM
    return fl


def gemm_entries(gprof, m_rows, fp8):
    """roofline entries of the three block GEMM shapes from the per-launch events (ops.PROFILE_GEMM): achieved = 2 M N K /
    mean launch time against the dense MFMA peak of the operand type; traffic = the static PMC figure of that shape (fabric-side
    bytes per launch) beside the algorithmic bytes A + W + C (+ residual) the launch must move at least."""
    peak = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS
    eb = 1 if fp8 else 2                                   # bytes per operand element
    out = []
    for key, n, kk, name, extra_c in (
            ("ffn1", FFN, DIM, "gemm_a4_kernel<GELU> FFN1 [S,5120]x[13824,5120]^T", 0),
            ("ffn2", DIM, FFN, "gemm_a4_kernel<gate*+resid> FFN2 [S,13824]x[5120,13824]^T", 1),
            ("dd", DIM, DIM, "gemm_a4_kernel D->D projections (q,k,v,o, zero-conv)", 0)):
        ms = [a.elapsed_time(b) for a, b, M, N, K, _ in gprof if (M, N, K) == (m_rows, n, kk)]
        if not ms:
            continue
        avg = sum(ms) / len(ms)
        fl = 2.0 * m_rows * n * kk
        traffic, src = static_traffic(("gemm_fp8_" if fp8 else "gemm_") + key)
        out.append({"bound": "mfma", "kernel": name + (" — e4m3 operands, v_mfma_f32_16x16x128_f8f6f4" if fp8 else ""),
                    "achieved": fl / (avg * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl / (avg * 1e-3) / 1e12 / peak,
                    "traffic": traffic, "traffic_static": True, "traffic_source": src,
                    "algorithmic_bytes_per_launch": m_rows * kk * eb + n * kk * eb + (1 + extra_c) * m_rows * n * 2,
                    "launches": len(ms), "avg_launch_ms": avg, "algorithmic_flops_per_launch": fl})
    return out


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(torch, runs=3):
    """Reference-equivalent CPU path (the oracle's torch-CPU restatement of DiTBlock, pinned to the
    reference by tests/test_oracle_goldens.py) on BASELINE config 1: one A14B block, S=14040, bf16, every host core.
    SURVEY §8(d): 1 untimed warm-up, then the median of 3 timed runs; CPU model and core count reported."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_inputs as gi
    from oracle import wan_oracle as wo
    cfg = gi.A14B
    s = 9 * 30 * 52
    sd = gi.block_sd(torch.Generator().manual_seed(31), cfg["dim"], cfg["ffn_dim"], "", torch.bfloat16)
    x, ctx, t_mod = gi.block_inputs(cfg["dim"], s, 512, seed=32)
    freqs = wo.rope_freqs_3d(128, 9, 30, 52)
    cores = torch.get_num_threads()
    times = []
    for i in range(runs + 1):
        t0 = time.time()
        wo.dit_block(x, ctx, t_mod, freqs, sd, "", cfg["num_heads"], cfg["eps"])
        if i:                      # run 0 is the warm-up (thread pool, allocator, page faults)
            times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    tflops = block_flops(s) / dt / 1e12
    loop_flops = 21 * 2 * forward_flops(True) + 29 * 2 * forward_flops(True)  # reference runs ControlNet2 too
    return {"value": 81.0 / (loop_flops / (tflops * 1e12)), "unit": "frames/s (derived: 50-step loop FLOPs / measured CPU FLOP/s)",
            "cores": cores, "cpu_model": cpu_model_name(), "kind": "port",
            "sample": f"one A14B DiTBlock fwd, S=14040 (BASELINE config 1), bf16, torch CPU eager, 1 warm-up + median of "
                      f"{runs}: {dt:.2f} s = {tflops:.2f} TFLOP/s (runs: {', '.join(f'{t:.2f}' for t in times)} s)",
            "block_seconds": dt, "tflops": tflops}


def conv_kernel_name(kt, ks, mode, c, n):
    """Which kernel gf_conv3d_bf16 sends a shape to (gf_gemm.hip::gf_conv3d_bf16, the same rules)."""
    if mode == 3:
        return "conv_a4_kernel (direct 3x3x3 on the 4-wave GEMM loop: zero-bordered activation, taps as row shifts, 256 x 192 tile)"
    if kt == 3 and ks == 3 and mode == 0 and c == 96 and (n == 96 or n <= 16):
        return "conv3d_c96_kernel (direct 3x3x3, halo staged once per frame, three temporal accumulator sets)"
    if kt == 1 and ks == 3 and mode == 1 and c == 192 and n == 96:
        return "conv2d_up_c192_kernel (direct 3x3 with the nearest-exact 2x upsample folded into the halo)"
    return "gemm_ph_kernel<CONV> (8-wave implicit GEMM, LDS-DMA gather, 256 x 192 tile at Cout = 192 / 384)"


def vae_entries(cprof, top=3):
    """roofline entries of the tiled VAE decode's dominant convolution launches (BASELINE config 4) from per-launch events
    (ops.PROFILE_CONV): launches are grouped by shape; per group algorithmic FLOPs = 2 rows Cout (kt ks^2 Cin), algorithmic bytes =
    input + weights + output (+ residual) in bf16, mean launch time, fraction of the dense bf16 MFMA peak.  The `top` groups by total
    time are returned with the share of the decode's convolution time each one is."""
    groups = {}
    for e0, e1, rows, n, c, kt, ks, mode, has_resid in cprof:
        groups.setdefault((rows, n, c, kt, ks, mode, has_resid), []).append(e0.elapsed_time(e1))
    total = sum(sum(v) for v in groups.values()) or 1.0
    out = []
    for (rows, n, c, kt, ks, mode, has_resid), ms in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:top]:
        taps = kt * ks * ks
        fl = 2.0 * rows * n * taps * c
        in_rows = rows // 4 if mode == 1 else (rows * 4 if mode == 2 else rows)      # the nearest-exact 2x upsample / stride-2 read
        by = in_rows * c * 2 + n * taps * c * 2 + rows * n * 2 * (2 if has_resid else 1)
        avg = sum(ms) / len(ms)
        kern = conv_kernel_name(kt, ks, mode, c, n)
        what = f"{kt}x{ks}x{ks} conv {c}->{n} over {rows} output pixels" + (" +residual" if has_resid else "") + \
               (" (2x upsample folded in)" if mode == 1 else " (stride 2)" if mode == 2 else "")      # (mode 3 = the padded layout: a few per cent more rows are read; the figures stay the algorithmic ones)
        traffic, src = static_traffic(f"vae_conv_c{c}_n{n}_k{kt}{ks}_m{mode}")
        out.append({"bound": "mfma", "kernel": kern, "launch": what, "achieved": fl / (avg * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": fl / (avg * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_static": True,
                    "traffic_source": src, "algorithmic_bytes_per_launch": by, "algorithmic_flops_per_launch": fl, "launches": len(ms),
                    "avg_launch_ms": avg, "share_of_decode_conv_time": sum(ms) / total})
    return out, total


def preloop_seconds(torch, vae, dev):
    """The units in front of the denoising loop at full size ("text / VAE encode reported separately"): two tiled VAE
    encodes of an 81-frame 480x832 clip (control video GF:791-805, image conditioning GF:887-917; tile (30,52)/(15,26)) and two
    umT5-XXL forwards at 512 tokens (positive + negative prompt, wan_video_text_encoder.py:209-255), random-init weights of the
    real shapes.  1 warm-up + 1 timed each."""
    from goal_force_amd.text_encoder import WanTextEncoder
    out = {}
    video = (torch.rand((3, 81, 480, 832), device=dev) * 2 - 1).to(torch.bfloat16)
    kw = dict(device=dev, tiled=True, tile_size=(30, 52), tile_stride=(15, 26))
    vae.encode([video], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z = vae.encode([video], **kw)
    z2 = vae.encode([video], **kw)
    torch.cuda.synchronize()
    out["vae_tiled_encode_x2_s"] = time.perf_counter() - t0
    assert tuple(z.shape) == (1, 16, 21, 60, 104) and bool(torch.isfinite(z.float()).all()) and torch.equal(z, z2)
    del video, z, z2
    with torch.device("meta"):
        te = WanTextEncoder()
    te = te.to_empty(device=dev).to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(1)
    for n, prm in te.named_parameters():
        if n.endswith("norm.weight") or "norm1" in n or "norm2" in n:
            prm.data.fill_(1.0)
        else:
            prm.data.copy_(torch.randn(prm.shape, generator=g, device=dev, dtype=torch.float32) * 0.02)
    ids = torch.randint(0, 256384, (1, 512), device=dev)
    mask = torch.zeros((1, 512), dtype=torch.long, device=dev)
    mask[:, :40] = 1
    te(ids, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e1 = te(ids, mask)
    e2 = te(ids, mask)
    torch.cuda.synchronize()
    out["umt5_xxl_512_tokens_x2_s"] = time.perf_counter() - t0
    assert bool(torch.isfinite(e1.float()).all()) and torch.equal(e1, e2)
    out["umt5_xxl_params"] = sum(prm.numel() for prm in te.parameters())
    out["total_s"] = out["vae_tiled_encode_x2_s"] + out["umt5_xxl_512_tokens_x2_s"]
    out["what"] = ("per video, in front of the loop: 2 tiled VAE encodes of [3,81,480,832] (control video, first-frame conditioning) + "
                   "2 umT5-XXL forwards at 512 tokens; random-init weights; NOT part of `value` (SURVEY §8d: reported separately)")
    return out


def gpu_eager_yardstick(torch, pipe, cfg, inp, log):
    """The SAME full-size step through torch-ROCm's own kernels — hipBLASLt GEMMs (F.linear), F.scaled_dot_product_attention (the
    reference's attention backend on this stack: flash_attn / sageattention are absent, DIT:55-60), eager LayerNorm / RMSNorm / RoPE /
    GELU (DIT:92-111, 206-210) — i.e. the arithmetic the reference itself would run on this very GPU, as the checker's bf16 graph
    (tests/fullsize_parity.py::OracleRunner over oracle/wan_oracle.py, pinned to the reference by the goldens).  A yardstick beside
    `cpu_baseline`, never part of `value`: 1 untimed warm-up step (library initialisation, autotuning), then one high-noise step
    (cond + uncond forward with the 10-block ControlNet, CFG, Euler), one low-noise step as the reference runs it (its all-zero
    ControlNet2 computed) and one with ControlNet2 elided as the product does (bit-identical)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from fullsize_parity import LazySD, OracleRunner
    run = OracleRunner(pipe, cfg, torch.bfloat16)
    wo = run.wo
    sigmas, timesteps = wo.flow_match_sigmas(50, 5.0)
    with_cn2 = (LazySD(pipe.dit2, torch.bfloat16), LazySD(pipe.controlnet2, torch.bfloat16), pipe.controlnet2.num_layers)

    def step(i, expert):
        old = run.experts[1]
        run.experts[1] = expert if expert is not None else old
        try:
            which = 0 if float(timesteps[i]) >= 875.0 else 1
            tsb = timesteps[i].unsqueeze(0).to(torch.bfloat16).to(inp["latents"].device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            posi = run.forward(which, inp["latents"], tsb, inp["ctx_p"], inp)
            nega = run.forward(which, inp["latents"], tsb, inp["ctx_n"], inp)
            lat = wo.euler_step(wo.cfg_combine(posi, nega, 5.0), i, inp["latents"], sigmas)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            run.experts[1] = old
        if not bool(torch.isfinite(lat.float()).all()):
            raise RuntimeError("non-finite latents in the eager yardstick")
        return dt * 1e3

    warm = step(12, None)
    hi = step(12, None)
    lo_ref = step(37, with_cn2)
    lo_elided = step(37, None)
    log(f"bench.py: eager yardstick (torch-ROCm kernels): warm-up {warm:.0f} ms, high-noise {hi:.0f} ms, low-noise {lo_ref:.0f} ms "
        f"(ControlNet2 run) / {lo_elided:.0f} ms (elided)")
    loop = (21 * hi + 29 * lo_ref) / 1e3
    return {"what": "one full-size denoise step of the same model, inputs and bf16 arithmetic through torch-ROCm's own kernels (hipBLASLt "
                    "F.linear, F.scaled_dot_product_attention, eager norms / RoPE / GELU): what the reference's code path runs on this GPU; "
                    "1 warm-up + 1 timed step of each kind; the denoising loop only (no VAE decode on either side of the ratio)",
            "denoise_step_ms_high_noise": hi, "denoise_step_ms_low_noise": lo_ref,
            "denoise_step_ms_low_noise_controlnet2_elided": lo_elided, "warmup_step_ms": warm,
            "ms_per_step": (21 * hi + 29 * lo_ref) / 50.0, "denoise_loop_s_50_steps": loop, "frames_per_sec": 81.0 / loop,
            "frames_per_sec_definition": "81 / (21 x high-noise + 29 x low-noise step as the reference runs it), denoise only",
            "torch": torch.__version__}


def tensor_digest(torch, t):
    """sha256 of the raw bytes + a few statistics of a device tensor (self-check of the timed run's outputs)."""
    import hashlib
    c = t.detach().contiguous().cpu()
    raw = c.view(torch.uint8) if c.dtype == torch.uint8 else c.view(torch.int16) if c.element_size() == 2 else c
    f = c.float()
    return {"sha256": hashlib.sha256(raw.numpy().tobytes()).hexdigest(), "mean": float(f.mean()), "std": float(f.std()),
            "absmax": float(f.abs().max())}


def host_threads_per_rank(n):
    return max(1, min(8, (os.cpu_count() or 8) // max(1, n)))


def self_launch(n):
    """`python bench.py --gpus N` with no launcher: N child processes of this file, one per GPU, torch.distributed.run's
    environment.  Runs in a parent that has imported neither torch nor anything else that initialises HIP (a process that has
    touched the GPU must not be replaced or forked on this pool); children are started fresh, never exec'd over the parent."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        # a free port now; if another process takes it before rank 0 binds it the ranks fail at rendezvous and the parent exits
        # non-zero with rank 0's message (set MASTER_PORT to pin one)
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
        sk.close()
    import signal
    procs = []

    def stop_children(*_):
        """terminate, then kill, every rank still alive (the parent must never leave ranks holding the GPUs behind)"""
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_end = time.time() + 10.0
        for p in live:
            try:
                p.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_signal(signum, _frame):
        stop_children()
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
            # N ranks x (all host cores) OpenMP threads each thrash the host: 8 ranks spent 70 s building a VAE that one rank builds in
            # 0.5 s (profiles/r06/bench_n8_phases.*).  torch.distributed.run sets OMP_NUM_THREADS=1 for the same reason; host work here is
            # weight initialisation and checksums only, so a few threads per rank are plenty.  The caller's own setting wins.
            env.setdefault("OMP_NUM_THREADS", str(host_threads_per_rank(n)))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else sys.stderr))
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:          # one rank failed: the others would wait in a collective forever
                    rc = code
                    print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in alive:
                        q.terminate()
            time.sleep(0.2)
    finally:
        stop_children()          # also reached when a Popen raised half-way or the poll loop was interrupted
    sys.exit(rc if rc >= 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="BASELINE config 5: block Linears on the fp8_linear contract (e4m3 MFMA)")
    ap.add_argument("--sp", type=int, default=1, help="head-parallel (Ulysses) degree inside each forward: world = videos x 2 x sp "
                    "(latency mode; the default 1 is the CFG-pair x sample layout the driver's scaling runs use)")
    ap.add_argument("--inputs", choices=["randn", "example"], default="randn",
                    help="conditioning of the timed run: seeded N(0,1) tensors (default, SURVEY §8d config 2) or the reference's pendulum "
                         "example (tests/golden/example_pendulum.npz): image conditioning y and force-map control latents are VAE "
                         "encodings of the real first frame and of the rendered goal-force map (the kernels' speed depends on the data)")
    ap.add_argument("--peaky", type=float, default=1.0, help="multiply every self-attention norm_q weight by this factor (8: attention "
                    "logits x 8, near-one-hot softmax rows) — data-sensitivity runs only")
    ap.add_argument("--peaky-steps", type=int, default=2, help="N = 1, bf16 run: after the timed region, time this many steps (1 warm-up first) with "
                    "every self-attention's logits x 8 (norm_q weights x 8: near-one-hot softmax rows, what trained attention looks like); "
                    "reported as `data_sensitivity`; 0 = skip")
    ap.add_argument("--no-preloop", action="store_true", help="skip the pre-loop timing (2 tiled VAE encodes + 2 umT5-XXL forwards)")
    ap.add_argument("--config5-steps", type=int, default=2, help="N = 1, bf16 run: after the bf16 timed region, switch the SAME modules to "
                    "the fp8_linear contract (BASELINE config 5) and time this many steps (1 warm-up first); reported as `config5` in the "
                    "JSON line; 0 = skip")
    ap.add_argument("--no-yardstick", action="store_true", help="skip the same-GPU eager yardstick (N = 1, bf16 run: the same step through "
                    "torch-ROCm's own kernels after the timed region, ~35 s; reported as `gpu_eager_yardstick`)")
    ap.add_argument("--no-launch-events", action="store_true", help="do not record the per-launch HIP events behind `roofline` / "
                    "`roofline_gemm` in the timed region (they are then null): measures what those ~900 event pairs per forward cost "
                    "(profiles/r03/README.md: below 0.1 %)")
    ap.add_argument("--layers", type=int, default=40, help=argparse.SUPPRESS)  # debugging only; 40 = the real model
    ap.add_argument("--sample-offset", type=int, default=0, help=argparse.SUPPRESS)  # tests: the N=1 run of video #k (seeds follow the sample id)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                  # never returns

    t_start = time.perf_counter()

    def phase(name):
        """wall-clock phases of the run on rank 0's stderr (never in the JSON line): where a launch spends its time outside the timed region"""
        if os.environ.get("RANK", "0") == "0":
            print(f"bench.py [{time.perf_counter() - t_start:7.2f} s] {name}", file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    from goal_force_amd import ops
    from goal_force_amd.distributed import CfgPairParallel, init_from_env
    from goal_force_amd.dit import A14B_CONFIG
    from goal_force_amd.pipeline import WanVideoPipeline, build_random_controlnet, build_random_expert

    torch.set_grad_enabled(False)
    phase("torch imported")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and "OMP_NUM_THREADS" not in os.environ:
        # ranks started by hand (tests) without a thread budget: the same budget self_launch() gives its children
        torch.set_num_threads(host_threads_per_rank(int(os.environ["WORLD_SIZE"])))
    rank, local, world = init_from_env()
    phase("process group ready")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.sp > 1 and world % (2 * args.sp):
        raise SystemExit(f"--sp {args.sp} needs --gpus = videos x 2 x sp")
    cfgp = CfgPairParallel(rank, world, sp_size=args.sp) if world > 1 else None
    seqp = None if cfgp is None else cfgp.sequence_parallel()
    sample = args.sample_offset + (0 if cfgp is None else cfgp.sample)

    cfg = dict(A14B_CONFIG)
    cfg["num_layers"] = args.layers
    n_cn = min(10, args.layers)
    dit = build_random_expert(cfg, seed=100, device=dev)
    dit2 = build_random_expert(cfg, seed=200, device=dev)
    cn = build_random_controlnet(n_cn, cfg, seed=300, device=dev)
    cn2 = build_random_controlnet(n_cn, cfg, seed=400, device=dev, zero_convs_zero=True)
    torch.cuda.synchronize()
    phase("experts and ControlNets built on the device")
    if args.fp8:
        from goal_force_amd.dit import enable_fp8
        for m in (dit, dit2, cn, cn2):
            enable_fp8(m)
    from goal_force_amd.vae import WanVideoVAE
    torch.manual_seed(7)
    vae = WanVideoVAE()                              # real Wan VAE architecture, random-init weights (CPU generator: the same on every rank)
    phase("VAE constructed on the host")
    vae = vae.to(torch.bfloat16).to(dev)
    pipe = WanVideoPipeline.from_modules(dit, dit2, cn, cn2, vae=vae, device=dev)
    torch.cuda.synchronize()
    phase("experts, ControlNets and VAE resident")

    # synthetic conditioning, per-video seed
    g = torch.Generator().manual_seed(1000 + sample)
    latents = pipe.generate_noise((1, 16, 21, 60, 104), seed=sample)
    y = torch.randn((1, 20, 21, 60, 104), generator=g)
    y[:, :4] = 0
    y[:, :4, 0] = 1  # first-frame mask (GF:898-909)
    y = y.to(torch.bfloat16).to(dev)
    control = torch.randn((1, 16, 21, 60, 104), generator=g).to(torch.bfloat16).to(dev)
    ctx_p = torch.randn((1, 512, 4096), generator=g)
    ctx_n = torch.randn((1, 512, 4096), generator=g)
    ctx_p[:, 40:] = 0
    ctx_n[:, 40:] = 0  # prompter zeroes past the prompt length (wan_prompter.py:99-109)
    ctx_p, ctx_n = ctx_p.to(torch.bfloat16).to(dev), ctx_n.to(torch.bfloat16).to(dev)
    if args.inputs == "example":
        # structured conditioning: the reference's pendulum example through the pipeline's own pre-loop units (random-init VAE)
        import numpy as np
        from PIL import Image
        from goal_force_amd.force_map import plan_control_video, render_control_video
        ex = np.load(os.path.join(ROOT, "tests", "golden", "example_pendulum.npz"))
        v = dict(zip([str(k) for k in ex["fields"]], [float(x) for x in ex["values"]]))
        masses = {"projectile": v["projectile_mass"], "target": v["target_mass"], "distractors": []}
        coords = {"projectile": [int(v["projectile_coordx"]), int(v["projectile_coordy"])],
                  "target": [int(v["target_coordx"]), int(v["target_coordy"])], "distractors": []}
        plan = plan_control_video(v["projectile_force_magnitude"], v["projectile_force_angle"], v["projectile_coordx"] / v["width"],
                                  v["projectile_coordy"] / v["height"], v["target_indirect_force_magnitude"],
                                  v["target_indirect_force_angle"], v["target_coordx"] / v["width"], v["target_coordy"] / v["height"],
                                  81, 480, 832, masses, coords, 30.0, 400.0, 30.0, 400.0, 1.0, 4.0, 0.0, 0.0, 0.0)   # INF:137-146
        cv = render_control_video(plan, dev)
        control = pipe.embed_control_video(cv, True, (30, 52), (15, 26))          # [1,16,21,60,104]
        y = pipe.embed_image(Image.fromarray(ex["image"]), 81, 480, 832, True, (30, 52), (15, 26))
        assert tuple(y.shape) == (1, 20, 21, 60, 104) and tuple(control.shape) == (1, 16, 21, 60, 104), (y.shape, control.shape)
    if args.peaky != 1.0:
        for m in (dit, dit2, cn, cn2):
            for blk in m.modules():
                if hasattr(blk, "self_attn") and hasattr(blk.self_attn, "norm_q"):
                    blk.self_attn.norm_q.weight.data.mul_(args.peaky)

    n_sched = 50
    k = max(1, args.steps)
    # K < 50: the K timed steps sit at the MIDPOINTS of K equal stretches of the schedule, so that their mix of high-noise steps
    # (21 of 50: both experts' ControlNets run) and low-noise steps follows the schedule (K = 5: 2 of 5, K = 10: 4 of 10);
    # K = 1 times a high-noise step (the pessimistic choice).  `frames_per_sec_schedule_weighted` is the exact 21/29 weighting.
    step_ids = (sorted({min(n_sched - 1, ((2 * i + 1) * n_sched) // (2 * k)) for i in range(k)}) if k > 1 else [0]) \
        if k < n_sched else list(range(n_sched))
    while len(step_ids) < k:  # K > 50: wrap around
        step_ids = step_ids + step_ids[: k - len(step_ids)]
    warm_ids = [step_ids[i % len(step_ids)] for i in range(args.warmup)]

    def run(ids, record=False):
        return pipe.denoise(latents, ctx_p, ctx_n, y, control, num_inference_steps=n_sched, cfg_scale=5.0,
                            controlnet=True, step_ids=ids, cfg_parallel=cfgp, record_step_times=record,
                            sequence_parallel=seqp)

    # N > 1: every collective of the run once at its production size, content-checked, BEFORE anything is timed — a transport
    # problem surfaces here with the backend's own message, not as a hang or a wrong number in the timed region
    pre = None
    if cfgp is not None:
        from goal_force_amd.distributed import preflight
        pre = preflight(cfgp, dev, log=(lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None)
        if rank == 0:
            print(f"bench.py pre-flight: backend {pre['backend']}, rccl_ranks {pre['rccl_ranks']}, min free HBM {pre['min_hbm_free_gb']:.1f} GB; "
                  + "; ".join(f"rank {r['rank']} -> {r['device']} (video {r['sample']}, branch {r['branch']}, sp {r['sp_rank']})"
                              for r in pre["ranks"]), file=sys.stderr, flush=True)
    phase("inputs built" + ("" if pre is None else ", pre-flight done"))
    if warm_ids:
        run(warm_ids)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    phase("warm-up done, timed region starts")
    ops.PROFILE_ATTN, ops.PROFILE_GEMM = (None, None) if args.no_launch_events else ([], [])
    t0 = time.perf_counter()
    final = run(step_ids, record=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    phase("timed region done")
    prof, ops.PROFILE_ATTN = ops.PROFILE_ATTN or [], None
    gprof, ops.PROFILE_GEMM = ops.PROFILE_GEMM or [], None
    step_ms = list(pipe.last_step_ms)           # (ms, is_low_noise) of the timed steps (the config-5 leg below records its own)
    # ---- self-check of what the timed steps produced: a forward that overflowed would time the same
    if not bool(torch.isfinite(final.float()).all()):
        raise SystemExit(f"rank {rank}: non-finite latents after the timed steps")
    lat_digest = tensor_digest(torch, final)
    if not (1e-3 < lat_digest["std"] < 1e3):
        raise SystemExit(f"rank {rank}: degenerate latents after the timed steps: {lat_digest}")
    # VAE tiled decode of THIS run's latents (GF:733; tile (30,52)/(15,26)), timed after the K steps: 1 warm-up + 1 timed
    # N > 1: the 9 tiles are split over the two ranks of the CFG pair (both hold the final latents) and exchanged
    tgroup = None if cfgp is None else cfgp.pair_group
    vae.decode(final, tiled=True, tile_size=(30, 52), tile_stride=(15, 26), tile_group=tgroup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    tv = time.perf_counter()
    frames = vae.decode(final, tiled=True, tile_size=(30, 52), tile_stride=(15, 26), tile_group=tgroup)
    torch.cuda.synchronize()
    vae_s = time.perf_counter() - tv
    assert tuple(frames.shape) == (1, 3, 81, 480, 832)
    if not bool(torch.isfinite(frames.float()).all()):
        raise SystemExit(f"rank {rank}: non-finite decoded frames")
    # per-launch events of the decode's convolutions (config 4's roofline entries): a third decode, outside `vae_s`
    vae_roofline, vae_conv_ms = None, None
    if world == 1 and not args.no_launch_events:
        ops.PROFILE_CONV = []
        vae.decode(final, tiled=True, tile_size=(30, 52), tile_stride=(15, 26))
        torch.cuda.synchronize()
        cprof, ops.PROFILE_CONV = ops.PROFILE_CONV, None
        vae_roofline, vae_conv_ms = vae_entries(cprof)
    u8 = pipe.frames_uint8(frames)                      # [81,480,832,3] uint8, 97 MB (what the reference saves, GF:735)
    frames_digest = tensor_digest(torch, u8)
    # ---- the host side of the B1 boundary (N = 1): `pipe(...)` is handed a host control-signal video ([81,480,832,3] bf16, 194 MB; the
    # reference's dataset builds it on the CPU) and an image, and returns 81 PIL frames (97 MB device -> host + the PIL wrapping, UTIL:85-91).
    # `value` is measured with inputs resident in HBM; this is what the PCIe crossings and the PIL conversion add per video.
    host_boundary = None
    if world == 1:
        try:
            hc = torch.zeros((81, 480, 832, 3), dtype=torch.bfloat16).pin_memory()
            hc.to(dev)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            hc.to(dev)
            torch.cuda.synchronize()
            h2d = time.perf_counter() - tb
            tb = time.perf_counter()
            host_u8 = u8.cpu()
            d2h = time.perf_counter() - tb
            tb = time.perf_counter()
            pil = pipe.vae_output_to_video(frames)
            pil_s = time.perf_counter() - tb
            assert len(pil) == 81 and pil[0].size == (832, 480) and tuple(host_u8.shape) == (81, 480, 832, 3)
            host_boundary = {"h2d_control_video_s": h2d, "h2d_bytes": hc.numel() * 2, "d2h_frames_s": d2h, "d2h_bytes": host_u8.numel(),
                             "frames_to_pil_s": pil_s, "what": "per video, outside `value`: pinned-host -> HBM copy of the control-signal video, HBM -> host "
                             "copy of the uint8 frames, and vae_output_to_video (uint8 conversion + D2H + 81 PIL images) as pipe(...) returns them"}
            del hc, host_u8, pil
        except Exception as e:      # noqa: BLE001
            host_boundary = {"error": f"{type(e).__name__}: {e}"}
    phase("VAE decodes and digests done")
    # end-of-run all-gather of every sample's frames among the samples' lead ranks (one contributor per sample, the
    # other ranks send nothing): 1 warm-up + 1 timed.  Rank 0 is a lead and ends up with every video's frames.
    gather_s, n_gathered = 0.0, 1
    sample_digests = [{"sample": sample, "latents_sha256": lat_digest["sha256"], "frames_uint8_sha256": frames_digest["sha256"]}]
    if cfgp is not None:
        cfgp.gather_frames(u8, tuple(u8.shape), torch.uint8, dev)
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        allf = cfgp.gather_frames(u8, tuple(u8.shape), torch.uint8, dev)
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - tg
        if cfgp.is_lead:
            n_gathered = len(allf)
            if not torch.equal(allf[cfgp.sample], u8):
                raise SystemExit(f"rank {rank}: the gathered frames of sample {cfgp.sample} differ from the local decode")
        # every video's checksums on rank 0: frames from the gathered tensors themselves, latents from their owners
        lat_all = [None] * world
        dist.all_gather_object(lat_all, (sample, cfgp.is_lead, lat_digest["sha256"]))
        if rank == 0:
            lat_by_sample = {sm: h for sm, lead, h in lat_all if lead}
            sample_digests = [{"sample": args.sample_offset + i, "latents_sha256": lat_by_sample[args.sample_offset + i],
                               "frames_uint8_sha256": tensor_digest(torch, f)["sha256"]} for i, f in enumerate(allf)]
    phase("frame all-gather done")
    if world > 1:
        t = torch.tensor([elapsed, vae_s, gather_s], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, vae_s, gather_s = (float(v) for v in t.tolist())

    # ---- BASELINE config 5 on the same modules (N = 1, bf16 run only): every block Linear on the fp8_linear contract
    # (VRAM:115-151), 1 warm-up + --config5-steps timed steps spread over the schedule like the bf16 ones
    config5 = None
    if world == 1 and not args.fp8 and args.config5_steps > 0:
        from goal_force_amd.dit import enable_fp8
        try:
            for m in (dit, dit2, cn, cn2):
                enable_fp8(m)
            k5 = args.config5_steps
            ids5 = sorted({min(n_sched - 1, ((2 * i + 1) * n_sched) // (2 * k5)) for i in range(k5)}) if k5 > 1 else [0]
            run([ids5[0]])
            torch.cuda.synchronize()
            ops.PROFILE_ATTN, ops.PROFILE_GEMM = (None, None) if args.no_launch_events else ([], [])
            t5 = time.perf_counter()
            final5 = run(ids5, record=True)
            torch.cuda.synchronize()
            el5 = time.perf_counter() - t5
            prof5, ops.PROFILE_ATTN = ops.PROFILE_ATTN or [], None
            gprof5, ops.PROFILE_GEMM = ops.PROFILE_GEMM or [], None
            if not bool(torch.isfinite(final5.float()).all()):
                raise RuntimeError("non-finite latents after the timed fp8 steps")
            hi5 = [ms for ms, low in pipe.last_step_ms if not low]
            lo5 = [ms for ms, low in pipe.last_step_ms if low]
            att5 = [a.elapsed_time(b) for a, b, sq, skv, _ in prof5 if sq == skv == S_TOK]
            sps5 = el5 / len(ids5)
            w5 = (21 * sum(hi5) / len(hi5) + 29 * sum(lo5) / len(lo5)) / 1e3 if hi5 and lo5 else n_sched * sps5
            config5 = {"what": "BASELINE config 5: the same run with every nn.Linear of the DiT / ControlNet blocks on the fp8_linear contract "
                               "(per-row dynamic activation scale, unit weight scale, OCP e4m3; diffsynth/vram_management/layers.py:115-151)",
                       "dtype": "fp8-e4m3 block Linears (bf16 elsewhere, fp32 accumulate)", "steps": len(ids5), "warmup": 1, "step_ids": ids5,
                       "ms_per_step": sps5 * 1e3, "denoise_step_ms_high_noise": sum(hi5) / len(hi5) if hi5 else None,
                       "denoise_step_ms_low_noise": sum(lo5) / len(lo5) if lo5 else None,
                       "frames_per_sec": 81.0 / (w5 + vae_s),
                       "frames_per_sec_definition": "81 / (21 high-noise + 29 low-noise step times + VAE decode): the schedule's own mix",
                       "self_attention_avg_launch_ms": sum(att5) / len(att5) if att5 else None,
                       "latents": tensor_digest(torch, final5),
                       "roofline_gemm": gemm_entries(gprof5, S_TOK, True)}
        except Exception as e:      # noqa: BLE001 — the bf16 measurement above must reach the JSON line whatever happens here
            config5 = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py: the config-5 leg failed ({config5['error']}); the bf16 line is printed without it", file=sys.stderr)
        finally:
            ops.PROFILE_ATTN, ops.PROFILE_GEMM = None, None
            for m in (dit, dit2, cn, cn2):
                enable_fp8(m, False)

    # ---- data sensitivity (N = 1, bf16): the same steps with every self-attention's logits x 8 — random-init attention is
    # near-uniform (the friendliest data for the lazy-rescale softmax); trained attention is peaky
    sensitivity = None
    if world == 1 and not args.fp8 and args.peaky == 1.0 and args.peaky_steps > 0:
        qn = [blk.self_attn.norm_q.weight for m in (dit, dit2, cn, cn2) for blk in m.modules()
              if hasattr(blk, "self_attn") and hasattr(blk.self_attn, "norm_q")]
        try:
            for w in qn:
                w.data.mul_(8.0)               # exact in bf16 (a power of two): undone exactly below
            kp = args.peaky_steps
            idsp = sorted({min(n_sched - 1, ((2 * i + 1) * n_sched) // (2 * kp)) for i in range(kp)}) if kp > 1 else [0]
            run([idsp[0]])
            torch.cuda.synchronize()
            ops.PROFILE_ATTN = None if args.no_launch_events else []
            finalp = run(idsp, record=True)
            torch.cuda.synchronize()
            profp, ops.PROFILE_ATTN = ops.PROFILE_ATTN or [], None
            hip_ = [ms for ms, low in pipe.last_step_ms if not low]
            lop_ = [ms for ms, low in pipe.last_step_ms if low]
            attp = [a.elapsed_time(b) for a, b, sq, skv, _ in profp if sq == skv == S_TOK]
            if not bool(torch.isfinite(finalp.float()).all()):
                raise RuntimeError("non-finite latents with the logits x 8")
            wp = (21 * sum(hip_) / len(hip_) + 29 * sum(lop_) / len(lop_)) / 1e3 if hip_ and lop_ else None
            sensitivity = {"what": "the same bf16 steps with every self-attention's logits x 8 (norm_q weights x 8): near-one-hot softmax rows, "
                                   "the rescale path of the attention kernel is taken as often as trained weights would take it",
                           "steps": len(idsp), "warmup": 1, "step_ids": idsp,
                           "denoise_step_ms_high_noise": sum(hip_) / len(hip_) if hip_ else None,
                           "denoise_step_ms_low_noise": sum(lop_) / len(lop_) if lop_ else None,
                           "frames_per_sec": 81.0 / (wp + vae_s) if wp else None,
                           "self_attention_avg_launch_ms": sum(attp) / len(attp) if attp else None}
        except Exception as e:      # noqa: BLE001
            sensitivity = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py: the data-sensitivity leg failed ({sensitivity['error']})", file=sys.stderr)
        finally:
            ops.PROFILE_ATTN = None
            for w in qn:
                w.data.mul_(0.125)

    # ---- pre-loop units at full size (N = 1): reported separately, never part of `value`
    preloop = None
    if world == 1 and not args.no_preloop:
        try:
            preloop = preloop_seconds(torch, vae, dev)
        except Exception as e:      # noqa: BLE001
            preloop = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py: the pre-loop leg failed ({preloop['error']})", file=sys.stderr)

    # ---- same-GPU yardstick (N = 1, bf16, full depth): the step through torch-ROCm's own kernels, reported beside cpu_baseline
    yardstick = None
    if world == 1 and not args.fp8 and not args.no_yardstick and args.peaky == 1.0:
        try:
            yardstick = gpu_eager_yardstick(torch, pipe, cfg, dict(latents=latents, y=y, control=control, ctx_p=ctx_p, ctx_n=ctx_n),
                                            lambda m: print(m, file=sys.stderr, flush=True))
        except Exception as e:      # noqa: BLE001 — the measurement above must reach the JSON line whatever happens here
            yardstick = {"error": f"{type(e).__name__}: {e}"}
            print(f"bench.py: the eager-yardstick leg failed ({yardstick['error']})", file=sys.stderr)
        torch.cuda.empty_cache()

    phase("extra legs done")
    if rank == 0:
        attn_traffic, attn_traffic_src = static_traffic("attn_self")
        sec_per_step = elapsed / k
        videos = 1 if world == 1 else world // (2 * args.sp)
        hi = [ms for ms, low in step_ms if not low]
        lo = [ms for ms, low in step_ms if low]
        # the 50-step loop: 21 high-noise + 29 low-noise steps (boundary 0.875 on the shift-5 schedule), each at its measured mean —
        # the K timed steps cannot hold that ratio exactly (K = 20: 8 + 12), so the headline weights them instead of scaling their sum
        if hi and lo and k < n_sched:
            loop_s = (21 * sum(hi) / len(hi) + 29 * sum(lo) / len(lo)) / 1e3
            loop_how = "21 x mean high-noise step + 29 x mean low-noise step (HIP events around each step on the launch stream)"
        else:
            loop_s = n_sched * sec_per_step
            loop_how = "all 50 steps timed" if k >= n_sched else "50 x the mean timed step (one kind of step timed only)"
        value = videos * 81.0 / (loop_s + vae_s + gather_s)
        # dominant kernel: self-attention flash-attention launches (q_len == kv_len == S)
        self_att = [(a.elapsed_time(b)) for a, b, sq, skv, _ in prof if sq == skv == S_TOK]
        att_ms = sum(self_att) / max(1, len(self_att))
        att_flops = 4.0 * S_TOK * S_TOK * DIM / args.sp   # a launch covers 40/sp heads under head parallelism
        achieved = att_flops / (att_ms * 1e-3) / 1e12 if self_att else None
        fwd_per_step = 2 if world == 1 else 1.0 / args.sp   # forwards computed by ONE rank per step
        n_hi = sum(1 for i in step_ids if i < 21)
        # executed FLOPs: with both CFG branches on one GPU the context-independent half of block 0 (self-attention + its four
        # projections; DiT, and ControlNet on the high-noise steps) is computed once per step, not twice (model_fn: cfg_shared)
        half0 = 4.0 * S_TOK * S_TOK * DIM + 8.0 * S_TOK * DIM * DIM
        saved = (lambda i: (2 if i < 21 else 1) * half0) if (world == 1 and pipe.share_cfg_prefix) else (lambda i: 0.0)
        step_flops = [(forward_flops(True) if i < 21 else forward_flops(False)) * fwd_per_step - saved(i) for i in step_ids]
        out = {
            "metric": "frames_per_sec (81-frame video / (50-step denoise loop + VAE decode), Wan2.2-I2V-A14B 832x480x81f)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": k, "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8-e4m3 block Linears (bf16 elsewhere, fp32 accumulate)" if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": "Goal-Force denoise step: cond+uncond model_fn (40 DiT + 10 ControlNet blocks, "
                                   "A14B dims) + CFG + Euler, latents [1,16,21,60,104] = 32760 tokens, 512 text tokens; "
                                   "random-init bf16 weights, both experts + both ControlNets resident",
                       "inputs": ("seeded N(0,1) conditioning tensors" if args.inputs == "randn" else
                                  "pendulum example of the reference (first frame + rendered goal-force map) through the pre-loop VAE encodes")
                                 + ("" if args.peaky == 1.0 else f"; self-attention logits x {args.peaky:g}"),
                       "schedule": f"FlowMatch 50 steps shift 5, boundary 0.875; timed step ids {step_ids} "
                                   f"({n_hi} high-noise with ControlNet, {k - n_hi} low-noise with the all-zero ControlNet2 elided)",
                       "layers": args.layers,
                       "cross_attention": "the context's padded rows (identical: the prompter zeroes past the 40-token prompt) are attended as "
                                          "ONE key with multiplicity 472 (41 keys instead of 512; ops.options(fold_pad_keys=False) attends all 512)"
                                          if ops._OPT["fold_pad_keys"] else "all 512 context keys attended",
                       "self_attention_q": "Q' = bf16(scale log2e x rotated q) written by the RMSNorm + RoPE kernel (one rounding; the attention runs with "
                                           "an in-kernel factor of exactly 1)" if ops._OPT["attn_q_prescale"] else
                                           "plain q, scaled and rounded again inside the attention kernel",
                       "parallelism": "1 GPU: sequential CFG (block 0's context-independent half shared by the two branches)" if world == 1 else f"{videos} video(s) x CFG pair, RCCL all-gather of noise_pred per step"
                                      + (f"; head-parallel attention degree {args.sp} (RCCL all-to-all over xGMI)" if args.sp > 1 else ""),
                       "value_definition": "frames/s = videos * 81 / (50-step loop + tiled VAE decode + frame all-gather); 50-step loop = " + loop_how,
                       "vae_decode": "tiled (30,52)/(15,26) decode of [1,16,21,60,104] on the HIP kernels, measured after the "
                                     "timed steps and included in value"
                                     + ("" if world == 1 else "; tiles split over the two ranks of each CFG pair")},
            "vae_decode_s": vae_s, "denoise_loop_s_50_steps": loop_s,
            "frame_allgather_s": gather_s, "samples_gathered": n_gathered,
            "distributed": {"world": world, "backend": dist.get_backend() if world > 1 else None,
                            "rccl_ranks": world if (world > 1 and dist.get_backend() == "nccl") else 0,
                            "collectives": None if world == 1 else "per step: all-gather of the 4.19 MB noise prediction inside each "
                                           "CFG pair; VAE tiles broadcast inside the pair; end of run: all-gather of the uint8 frames (97 MB "
                                           "per sample) among the samples' lead ranks, timed as frame_allgather_s and included in value"},
            "self_check": {"latents_finite": True, "frames_finite": True, "latents": lat_digest, "frames_uint8": frames_digest,
                           "per_sample": sample_digests,
                           "note": "latents = output of the K timed steps of rank 0's sample (finite, std in (1e-3,1e3) or the run "
                                   "aborts); frames = tiled VAE decode of those latents -> uint8 as the reference saves them"},
            "frames_per_sec_denoise_only": videos * 81.0 / loop_s,
            "denoise_step_ms_high_noise": sum(hi) / len(hi) if hi else None,
            "denoise_step_ms_low_noise": sum(lo) / len(lo) if lo else None,
            "step_mfma_frac": (sum(step_flops) / elapsed / 1e12) / PEAK_BF16_TFLOPS,
            "frames_per_sec_unweighted": videos * 81.0 / (n_sched * sec_per_step + vae_s + gather_s),
            "roofline": {"bound": "mfma", "kernel": "flash_attn_fwd_kernel3<2> (self-attention, S=32760, 40 heads, d=128; V^T is written by the V projection GEMM, gf_linear_vt32)",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "peak_note": "dense bf16 MFMA peak = 256 CU x 4096 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md); under this "
                                      "load the chip holds 1.74-2.05 GHz depending on the box and build (rocprofv3 GRBM_GUI_ACTIVE: 1.84 GHz "
                                      "at 75 % MFMA-busy in profiles/r03/pmc/attn_SQ.md: 1.87 GHz), i.e. 1.8-2.1 PFLOP/s",
                         "frac": None if achieved is None else achieved / PEAK_BF16_TFLOPS,
                         # HBM bytes per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes over
                         # tools/microbench.py attn, profiles/r01/pmc/): (2 x FETCH_SIZE [gfx950 reports half of a 16-B/lane
                         # stream] + WRITE_SIZE) KiB -> bytes.  Not collected live: PMC needs the profiler.
                         # The launch = flash_attn_fwd_kernel3<2> alone (V^T is written by the V projection, gf_linear_vt32).
                         "traffic": attn_traffic, "traffic_static": True,
                         "traffic_source": attn_traffic_src,
                         "algorithmic_bytes_per_launch": 4 * S_TOK * DIM * 2 // args.sp,      # Q + K + V read, O written
                         "launches": len(self_att), "avg_launch_ms": att_ms if self_att else None,
                         "algorithmic_flops_per_launch": att_flops},
        }
        # second entry: the FFN GEMMs (D->F with the GELU epilogue, F->D with gate*+residual) — with the four D->D
        # projections the GEMMs are the other ~45 % of a step
        out["roofline_gemm"] = gemm_entries(gprof, S_TOK // args.sp, args.fp8)
        if vae_roofline is not None:
            out["roofline_vae"] = vae_roofline
            out["vae_decode_conv_ms"] = vae_conv_ms
        if config5 is not None:
            out["config5"] = config5
        if sensitivity is not None:
            out["data_sensitivity"] = sensitivity
        if preloop is not None:
            out["preloop"] = preloop
        if host_boundary is not None:
            if "error" not in host_boundary:
                extra_s = host_boundary["h2d_control_video_s"] + host_boundary["frames_to_pil_s"]
                host_boundary["frames_per_sec_pcie_inclusive"] = videos * 81.0 / (loop_s + vae_s + gather_s + extra_s)
            out["host_boundary"] = host_boundary
        if yardstick is not None:
            if "error" not in yardstick:
                hi_ms, lo_ms = out["denoise_step_ms_high_noise"], out["denoise_step_ms_low_noise"]
                yardstick["speedup_denoise_loop"] = yardstick["denoise_loop_s_50_steps"] / loop_s
                yardstick["speedup_high_noise_step"] = yardstick["denoise_step_ms_high_noise"] / hi_ms if hi_ms else None
                yardstick["speedup_low_noise_step"] = yardstick["denoise_step_ms_low_noise"] / lo_ms if lo_ms else None
            out["gpu_eager_yardstick"] = yardstick
        if pre is not None:
            out["preflight"] = {"steps": pre["steps"], "rccl_ranks": pre["rccl_ranks"], "min_hbm_free_gb": pre["min_hbm_free_gb"],
                                "ranks": [{k_: r[k_] for k_ in ("rank", "sample", "branch", "sp_rank", "device", "hbm_free_gb")} for r in pre["ranks"]]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(torch)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
