"""Production-size ControlNet TRAINING-STEP parity tool (not collected by pytest): loss and every ControlNet gradient of one
`training_loss` + backward (GF:180-193; src/goal_force/utils.py:797-812) at the shape scripts/train/train_goal_force.sh trains on
— Wan2.2 A14B expert (40 blocks, frozen) + 10-block ControlNet, 81 frames of 832 x 480 = 32760 tokens — on the HIP path against
the REFERENCE'S ARITHMETIC at the same size: oracle/train_oracle.py (pinned to the reference's own training step by g9 on the
tiny configuration) under torch autograd on this GPU, in bf16 — what the reference computes here — and in fp32, the yardstick
(on the bf16-rounded timestep: GF:184 rounds the drawn timestep to the model dtype, and a yardstick on the unrounded one would be
a different training step — other sigma, other loss weight — 0.19 away from both bf16 runs whatever their arithmetic).

Autograd over 50 blocks at S = 32760 does not fit without recomputation (one block keeps ~20 GB of fp32 activations, the fp32
attention probabilities of one layer are 171 GB), so the oracle's blocks run under torch.utils.checkpoint (per block, and per
2048-query chunk of the fp32 attention): recomputation repeats the same ops on the same inputs and changes no value.  The
reference trains with gradient checkpointing too (train_goal_force.sh: --use_gradient_checkpointing).

    python tests/fullsize_train_parity.py                                     # 40 + 10 blocks, 21 latent frames (~5 GPU-minutes)
    python tests/fullsize_train_parity.py --layers 4 --cn-layers 2 --frames 5    # quick look

`--peaky f`: every self-attention's norm_q weight x f (logits x f).  The report also carries the HIP step under
ops.options(attn_q_prescale=False) — rounds 1-5's training path, whose forward rounds Q a second time.

Reported: loss (hip / reference bf16 / fp32); relative L2 distance from the fp32 gradients of hip and of the reference's bf16
arithmetic, over all ControlNet parameters together and per group (patch embedding, each block, zero-convs)."""
import argparse
import contextlib
import json
import math
import os
import sys
import time

import torch
from torch.utils.checkpoint import checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import train_oracle as to        # noqa: E402
from oracle import wan_oracle as wo          # noqa: E402

BF = torch.bfloat16
TEXT_TOKENS = 512


@contextlib.contextmanager
def recomputing(q_chunk):
    """oracle.wan_oracle with every dit_block under checkpoint and, for fp32 tensors, the attention in checkpointed query chunks
    (the same matmul / softmax / matmul as wan_oracle.attention_chunked)."""
    old = (wo.dit_block, wo.attention_chunked, wo.ATTENTION_Q_CHUNK)
    block = wo.dit_block

    def dit_block(x, context, t_mod, freqs, sd, pre, num_heads, eps=1e-6):
        return checkpoint(lambda x_, c_, t_: block(x_, c_, t_, freqs, sd, pre, num_heads, eps), x, context, t_mod, use_reentrant=False)

    def attention_chunked(q, k, v, num_heads, chunk):
        b, sq, hd = q.shape
        d = hd // num_heads

        def one(qc, k_, v_):
            kh = k_.reshape(b, k_.shape[1], num_heads, d).permute(0, 2, 3, 1)
            vh = v_.reshape(b, v_.shape[1], num_heads, d).transpose(1, 2)
            qh = qc.reshape(b, -1, num_heads, d).transpose(1, 2)
            p = torch.softmax((qh @ kh) * (1.0 / math.sqrt(d)), dim=-1)
            return (p @ vh).transpose(1, 2).reshape(b, -1, hd)
        return torch.cat([checkpoint(one, q[:, s:s + chunk], k, v, use_reentrant=False) for s in range(0, sq, chunk)], dim=1)

    wo.dit_block, wo.attention_chunked, wo.ATTENTION_Q_CHUNK = dit_block, attention_chunked, q_chunk
    try:
        yield
    finally:
        wo.dit_block, wo.attention_chunked, wo.ATTENTION_Q_CHUNK = old


def group_of(name):
    if name.startswith("controlnet_dit.blocks."):
        return "block " + name.split(".")[2]
    if name.startswith("controlnet_zero_convs_after"):
        return "zero-convs"
    return "patch embedding"


def distances(grads, ref):
    """Relative L2 distance of `grads` from `ref` (dicts name -> tensor): overall and per group, accumulated in fp64."""
    num, den = {}, {}
    for n, r in ref.items():
        g = grads[n]
        a = float((g.double() - r.double()).pow(2).sum())
        b = float(r.double().pow(2).sum())
        for key in ("all", group_of(n)):
            num[key] = num.get(key, 0.0) + a
            den[key] = den.get(key, 0.0) + b
    return {k: math.sqrt(num[k] / max(den[k], 1e-300)) for k in num}


def run(layers=40, cn_layers=10, frames=21, timestep_id=500, q_chunk=2048, peaky=1.0, log=print):
    from fullsize_parity import LazySD, config_of
    from goal_force_amd import training as tr
    from goal_force_amd.dit import A14B_CONFIG
    from goal_force_amd.pipeline import WanVideoPipeline, build_random_controlnet, build_random_expert
    dev = torch.device("cuda", 0)
    cfg = dict(A14B_CONFIG)
    cfg["num_layers"] = layers
    dit = build_random_expert(cfg, seed=100, device=dev)
    for p in dit.parameters():
        p.requires_grad_(False)
    cn = build_random_controlnet(cn_layers, cfg, seed=300, device=dev)
    if peaky != 1.0:                                                     # every self-attention's logits x peaky (what trained attention looks like)
        for m in (dit, cn):
            for n, p in m.named_parameters():
                if n.endswith("self_attn.norm_q.weight"):
                    p.data.mul_(peaky)
    pipe = WanVideoPipeline.from_modules(dit, None, cn, None, device=dev)
    pipe.scheduler.set_timesteps(1000, training=True)                    # utils.py:560
    g = torch.Generator().manual_seed(0)
    shp = (1, 16, frames, 60, 104)
    inp = dict(input_latents=torch.randn(shp, generator=g), noise=torch.randn(shp, generator=g), y=torch.randn((1, 20) + shp[2:], generator=g),
               control=torch.randn(shp, generator=g), context=torch.randn((1, TEXT_TOKENS, 4096), generator=g))
    inp = {k: v.to(BF).to(dev) for k, v in inp.items()}
    rep = {"config": {"layers": layers, "controlnet_layers": cn_layers, "tokens": frames * 30 * 52, "latents": list(shp), "timestep_id": timestep_id, "self_attention_logits_x": peaky,
                      "trainable_params": sum(p.numel() for p in cn.parameters()), "weights": "random init (bench.py's seeds)"}}

    # ---- the product: training_loss + backward through the HIP kernels
    def hip_step():
        for p in cn.parameters():
            p.grad = None
        with torch.enable_grad():
            loss = tr.training_loss(pipe, input_latents=inp["input_latents"], noise=inp["noise"], context=inp["context"], y=inp["y"],
                                    control_signal_video_latents=inp["control"], timestep_id=timestep_id)
            loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach())
    hip_step()
    t0 = time.time()
    hip_loss = hip_step()
    t_hip = time.time() - t0
    hip_grads = {n: p.grad.detach().clone() for n, p in cn.named_parameters()}
    assert all(p.grad is None for p in dit.parameters())
    from goal_force_amd import ops
    with ops.options(attn_q_prescale=False):                             # rounds 1-5's training path: a plain q, rounded again inside the forward kernel
        plain_loss = hip_step()
    plain_grads = {n: p.grad.detach().float().cpu() for n, p in cn.named_parameters()}
    for p in cn.parameters():
        p.grad = None
    log(f"hip: loss {hip_loss:.6f}, forward + backward {t_hip:.2f} s, peak {torch.cuda.max_memory_allocated() / 2 ** 30:.0f} GB")

    # ---- the reference's arithmetic under autograd, block-wise recomputation
    ocfg = config_of(dit)

    def oracle(dtype):
        dsd = LazySD(dit, dtype)
        csd = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in cn.state_dict().items()}
        i = {k: v.to(dtype) for k, v in inp.items()}
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.enable_grad(), recomputing(q_chunk):
            loss = to.training_loss(dsd, csd, ocfg, cn_layers, i["input_latents"], i["noise"], i["context"], i["y"], i["control"], timestep_id,
                                    timestep_dtype=BF)           # the fp32 yardstick sees the step the bf16 runs see (GF:184 rounds the timestep)
            loss.backward()
        torch.cuda.synchronize()
        log(f"oracle[{str(dtype).split('.')[-1]}]: loss {float(loss.detach()):.6f}, forward + backward {time.time() - t0:.1f} s")
        return float(loss.detach()), {k: v.grad for k, v in csd.items()}, time.time() - t0

    ref_loss, ref_grads, t_ref = oracle(BF)
    ref_grads = {k: v.float().cpu() for k, v in ref_grads.items()}       # make room for the fp32 run
    torch.cuda.empty_cache()
    f32_loss, f32_grads, t_f32 = oracle(torch.float32)
    f32_grads = {k: v.cpu() for k, v in f32_grads.items()}
    hip_cpu = {k: v.float().cpu() for k, v in hip_grads.items()}
    assert sorted(hip_cpu) == sorted(f32_grads)
    rep["loss"] = {"hip": hip_loss, "hip_plain_q": plain_loss, "ref_bf16": ref_loss, "fp32": f32_loss}
    rep["grad_rel_l2"] = {"hip_bf16_vs_fp32": distances(hip_cpu, f32_grads), "ref_bf16_vs_fp32": distances(ref_grads, f32_grads),
                          "hip_bf16_vs_ref_bf16": distances(hip_cpu, ref_grads), "hip_plain_q_vs_fp32": distances(plain_grads, f32_grads)}
    rep["grad_norm"] = {"hip": math.sqrt(sum(float(v.double().pow(2).sum()) for v in hip_cpu.values())),
                        "ref_bf16": math.sqrt(sum(float(v.double().pow(2).sum()) for v in ref_grads.values())),
                        "fp32": math.sqrt(sum(float(v.double().pow(2).sum()) for v in f32_grads.values()))}
    rep["seconds_forward_backward"] = {"hip": t_hip, "reference_arithmetic_on_torch_rocm_bf16_recomputing": t_ref, "fp32_recomputing": t_f32}
    log(f"  loss {rep['loss']}")
    log(f"  gradient norm {rep['grad_norm']}")
    for k in ("hip_bf16_vs_fp32", "ref_bf16_vs_fp32", "hip_bf16_vs_ref_bf16", "hip_plain_q_vs_fp32"):
        log(f"  {k}: " + ", ".join(f"{g} {v:.3e}" for g, v in rep["grad_rel_l2"][k].items()))
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--cn-layers", type=int, default=10)
    ap.add_argument("--frames", type=int, default=21, help="latent frames (21 = 81 video frames)")
    ap.add_argument("--timestep-id", type=int, default=500)
    ap.add_argument("--q-chunk", type=int, default=2048)
    ap.add_argument("--peaky", type=float, default=1.0, help="multiply every self-attention's norm_q weight by this factor (logits x f)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rep = run(a.layers, a.cn_layers, a.frames, a.timestep_id, a.q_chunk, a.peaky)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(rep, f, indent=1)
