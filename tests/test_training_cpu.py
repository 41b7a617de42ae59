"""Training-step oracle against the reference-generated golden (tests/golden/g9_training.npz: the reference's scheduler,
model_fn_wan_video and modules run through training_loss + loss.backward(), GF:180-193)."""
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN
from oracle import train_oracle as to


@pytest.fixture(autouse=True)
def _grad_enabled():
    """Other test modules switch autograd off process-wide at import; the training tests need the tape."""
    with torch.enable_grad():
        yield


def _g():
    return np.load(os.path.join(GOLDEN, "g9_training.npz"))


def test_training_schedule_matches_reference():
    g = _g()
    sig, ts, w = to.training_schedule(1000, 5.0)
    ids = g["sched_ids"]
    assert np.array_equal(sig[ids].numpy(), g["sched_sigmas"])
    assert np.array_equal(ts[ids].numpy(), g["sched_timesteps"])
    assert np.allclose(w[ids].numpy(), g["sched_weights"], rtol=1e-6, atol=0)


def test_product_scheduler_training_mode_matches_reference():
    from goal_force_amd.scheduler import FlowMatchScheduler
    g = _g()
    sch = FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(1000, training=True)
    ids = g["sched_ids"]
    assert np.array_equal(sch.sigmas[ids].numpy(), g["sched_sigmas"])
    assert np.array_equal(sch.timesteps[ids].numpy(), g["sched_timesteps"])
    assert np.allclose(sch.linear_timesteps_weights[ids].numpy(), g["sched_weights"], rtol=1e-6, atol=0)
    # training_weight looks the bf16-rounded timestep up again by nearest neighbour (FM:108-111): id 137 -> 968.0
    t = sch.timesteps[137:138].to(torch.bfloat16)
    sig, ts, w = to.training_schedule(1000, 5.0)
    tid = int(torch.argmin((ts - t.float()).abs()))
    assert float(sch.training_weight(t)) == float(w[tid])
    assert float(sch.add_noise(torch.zeros(1), torch.ones(1), t)) == float(sig[tid])


def _check(grads, g, mode, rtol):
    names = [str(n) for n in g["names"]]
    assert sorted(grads) == names
    for i, n in enumerate(names):
        gf = grads[n].double().flatten()
        nrm = float(g[f"norm_{mode}"][i])
        assert abs(float(gf.norm()) - nrm) <= rtol * nrm + 1e-12, n
        pr = gi.grad_probes(gf.numel(), seed=1000 + i) @ gf
        scale = nrm * np.sqrt(gf.numel()) + 1e-30                      # std of a projection of a vector of that norm
        assert float((pr - torch.from_numpy(g[f"proj_{mode}"][i])).abs().max()) <= rtol * scale, n
        smp = gf[gi.grad_sample_index(gf.numel(), seed=2000 + i)]
        ref = torch.from_numpy(g[f"sample_{mode}"][i])
        assert float((smp - ref).norm()) <= rtol * float(ref.norm()) + 1e-12, n


def test_oracle_training_step_fp32_matches_reference():
    g = _g()
    inp = gi.train_inputs()
    assert gi.same_checksum(gi.checksum(inp), g["ck_inputs"])
    cfg = gi.TINY
    loss, grads = to.loss_and_grads(gi.dit_sd(cfg, seed=41), gi.controlnet_sd(cfg, gi.TINY_CONTROLNET_LAYERS, seed=42), cfg,
                                    gi.TINY_CONTROLNET_LAYERS, inp, gi.TRAIN_TIMESTEP_ID, dtype=torch.float32)
    assert abs(float(loss) - float(g["loss_f32"])) < 1e-5 * float(g["loss_f32"])
    _check(grads, g, "f32", rtol=2e-4)


def test_oracle_training_step_bf16_matches_reference():
    g = _g()
    cfg = gi.TINY
    loss, grads = to.loss_and_grads(gi.dit_sd(cfg, seed=41), gi.controlnet_sd(cfg, gi.TINY_CONTROLNET_LAYERS, seed=42), cfg,
                                    gi.TINY_CONTROLNET_LAYERS, gi.train_inputs(), gi.TRAIN_TIMESTEP_ID, dtype=torch.bfloat16)
    assert abs(float(loss) - float(g["loss_bf16"])) < 2e-2 * float(g["loss_bf16"])
    _check(grads, g, "bf16", rtol=5e-2)       # same ops, same dtype; the oracle's LayerNorm is the fp32-autocast variant


def test_the_production_size_tool_recomputes_the_pinned_oracle_and_changes_no_value():
    """tests/fullsize_train_parity.py runs the oracle's blocks under torch.utils.checkpoint (and its fp32 attention in checkpointed query
    chunks) because autograd over 50 blocks at 32760 tokens does not fit otherwise: on the tiny configuration that form gives the
    gradients of the g9-pinned oracle — bit for bit in bf16 (SDPA either way), to summation order in fp32 (chunked attention)."""
    import fullsize_train_parity as ft
    cfg, n_cn = gi.TINY, gi.TINY_CONTROLNET_LAYERS
    dsd, csd0, inp = gi.dit_sd(cfg, seed=41), gi.controlnet_sd(cfg, n_cn, seed=42), gi.train_inputs()
    for dtype, bar in ((torch.bfloat16, 0.0), (torch.float32, 1e-5)):
        l0, g0 = to.loss_and_grads(dsd, csd0, cfg, n_cn, inp, gi.TRAIN_TIMESTEP_ID, dtype=dtype)
        d = {k: v.to(dtype) for k, v in dsd.items()}
        c = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in csd0.items()}
        i = {k: v.to(dtype) for k, v in inp.items()}
        with ft.recomputing(8):
            loss = to.training_loss(d, c, cfg, n_cn, i["input_latents"], i["noise"], i["context"], i["y"], i["control"], gi.TRAIN_TIMESTEP_ID)
            loss.backward()
        dist = ft.distances({k: v.grad for k, v in c.items()}, g0)
        assert abs(float(loss.detach()) - float(l0)) <= bar * abs(float(l0)) and dist["all"] <= bar, (dtype, dist)
        assert set(dist) == {"all", "patch embedding", "zero-convs"} | {f"block {b}" for b in range(n_cn)}


def test_an_fp32_yardstick_of_a_bf16_step_sees_the_bf16_rounded_timestep():
    """GF:184 rounds the drawn timestep to the model dtype; the sigma and the loss weight follow the ROUNDED value (FM:97, 109).  With
    timestep_dtype=bfloat16 an fp32 run of training_loss is the same training step as the bf16 run (close losses); without it, it
    is a neighbouring step wherever the rounding crosses a grid point."""
    cfg, n_cn = gi.TINY, gi.TINY_CONTROLNET_LAYERS
    dsd, csd, inp = gi.dit_sd(cfg, seed=41), gi.controlnet_sd(cfg, n_cn, seed=42), gi.train_inputs()
    _, ts, _ = to.training_schedule(1000, 5.0)
    moved = [t for t in range(1000) if int(torch.argmin((ts - ts[t].to(torch.bfloat16).float()).abs())) != t]
    assert moved, "bf16 rounding must move some timestep to a neighbouring grid point"
    tid = moved[len(moved) // 2]

    def loss(dtype, **kw):
        d = {k: v.to(dtype) for k, v in dsd.items()}
        c = {k: v.to(dtype) for k, v in csd.items()}
        i = {k: v.to(dtype) for k, v in inp.items()}
        with torch.no_grad():
            return float(to.training_loss(d, c, cfg, n_cn, i["input_latents"], i["noise"], i["context"], i["y"], i["control"], tid, **kw))
    lb, l32, l32r = loss(torch.bfloat16), loss(torch.float32), loss(torch.float32, timestep_dtype=torch.bfloat16)
    assert abs(l32r - lb) < abs(l32 - lb), (lb, l32, l32r)
    assert abs(l32r - lb) < 2e-2 * abs(lb), (lb, l32r)
