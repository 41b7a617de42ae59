"""CPU oracle of the ControlNet training step — TEST INFRASTRUCTURE ONLY (tests/ and tools may import it; the product path
under goal_force_amd/ never does).

Restates `WanVideoPipeline.training_loss` (src/goal_force/wan_video_new.py:180-193) and the scheduler's training mode
(diffsynth/schedulers/flow_match.py:34-67, 94-111) on top of the oracle's functional model_fn (oracle/wan_oracle.py, pinned
to the reference's model_fn by tests/golden/g5_model_fn.npz); gradients come from torch autograd over those same torch ops,
i.e. the arithmetic `accelerator.backward(loss)` runs in the reference (src/goal_force/utils.py:803).
Pinned by tests/golden/g9_training.npz, which tests/golden/make_goldens.py produced by running the REFERENCE's scheduler,
model_fn_wan_video and modules through the same eight lines.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import wan_oracle as wo


def training_schedule(num_steps: int = 1000, shift: float = 5.0):
    """FlowMatchScheduler(shift=5, sigma_min=0, extra_one_step=True).set_timesteps(num_steps, training=True)
    -> (sigmas, timesteps, linear_timesteps_weights), fp32 (FM:34-67)."""
    sigmas, timesteps = wo.flow_match_sigmas(num_steps, shift=shift)
    x = timesteps
    y = torch.exp(-2 * ((x - num_steps / 2) / num_steps) ** 2)
    y_shifted = y - y.min()
    return sigmas, timesteps, y_shifted * (num_steps / y_shifted.sum())


def training_loss(dit_sd, cn_sd, cfg, n_cn, input_latents, noise, context, y, control_latents, timestep_id,
                  num_steps: int = 1000, shift: float = 5.0, timestep_dtype=None):
    """GF:180-193 with the random timestep draw pinned to `timestep_id`.  Tensors carry the dtype to compute in.
    timestep_dtype: the dtype the drawn timestep is rounded to when it differs from the compute dtype — an fp32 YARDSTICK of a bf16
    run must see the bf16-rounded timestep (and the sigma / weight it selects), or it measures a different step (GF:184)."""
    dt = input_latents.dtype
    sigmas, timesteps, weights = training_schedule(num_steps, shift)
    timestep = timesteps[timestep_id:timestep_id + 1].to(timestep_dtype or dt).to(dt)   # GF:184 (rounded to the model dtype)
    tid = int(torch.argmin((timesteps - timestep.float()).abs()))                   # FM:97, 109
    timestep = timestep.to(input_latents.device)                                    # the inputs' device (tests/fullsize_train_parity.py runs this on the GPU)
    sigma = sigmas[tid]
    latents = (1 - sigma) * input_latents + sigma * noise                            # FM:94-100
    target = noise - input_latents                                                   # FM:103-105
    pred = wo.model_fn(dit_sd, cfg, latents, timestep, context, y=y, controlnet_sd=cn_sd, control_latents=control_latents,
                       num_controlnet_layers=n_cn)
    loss = F.mse_loss(pred.float(), target.float())
    return loss * weights[tid]


def loss_and_grads(dit_sd, cn_sd, cfg, n_cn, inputs, timestep_id, dtype=torch.float32, num_steps=1000, shift=5.0):
    """Loss and d(loss)/d(ControlNet parameter) for every entry of cn_sd; inputs: dict with input_latents, noise, context,
    y, control.  The DiT state dict is frozen, as in training (only `pipe.controlnet.*` is trainable, GF:97-117)."""
    with torch.enable_grad():
        return _loss_and_grads(dit_sd, cn_sd, cfg, n_cn, inputs, timestep_id, dtype, num_steps, shift)


def _loss_and_grads(dit_sd, cn_sd, cfg, n_cn, inputs, timestep_id, dtype, num_steps, shift):
    dsd = {k: v.detach().to(dtype) for k, v in dit_sd.items()}
    csd = {k: v.detach().to(dtype).requires_grad_(True) for k, v in cn_sd.items()}
    inp = {k: v.to(dtype) for k, v in inputs.items()}
    loss = training_loss(dsd, csd, cfg, n_cn, inp["input_latents"], inp["noise"], inp["context"], inp["y"], inp["control"],
                         timestep_id, num_steps, shift)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in csd.items()}
