"""FlowMatchScheduler — the flow-matching schedule Goal Force samples and trains on.

Interface of diffsynth/schedulers/flow_match.py ("FM") as Goal Force uses it: `FlowMatchScheduler(shift=5,
sigma_min=0.0, extra_one_step=True)` (GF:127), `set_timesteps(50, denoising_strength, shift=5.0)` (GF:663),
`set_timesteps(1000, training=True)` (utils.py:560), `step` (GF:721), `add_noise` / `training_target` /
`training_weight` (GF:186-192), attributes `.sigmas`, `.timesteps`, `.num_train_timesteps`, `.training`.

Only that configuration is built: the linear sigma ramp with the rational `shift` warp.  The reference's other
modes (exponential shift, terminal shift, reversed / inverse tables, `return_to_timestep`) are never reached by the
Goal-Force scripts; their constructor arguments are accepted and REFUSED with NotImplementedError.

Tables are host-side fp32 tensors, bit-exact with the reference (tests/golden/g1_scheduler.npz); the update on
device tensors is the gf_cfg_euler_step kernel.
"""
from __future__ import annotations

import torch

from . import ops

_UNBUILT = ("inverse_timesteps", "reverse_sigmas", "exponential_shift", "exponential_shift_mu", "shift_terminal")


class FlowMatchScheduler:
    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False,
                 exponential_shift=False, exponential_shift_mu=None, shift_terminal=None):
        given = dict(inverse_timesteps=inverse_timesteps, reverse_sigmas=reverse_sigmas,
                     exponential_shift=exponential_shift, exponential_shift_mu=exponential_shift_mu,
                     shift_terminal=shift_terminal)
        for name in _UNBUILT:
            if given[name]:
                raise NotImplementedError(f"FlowMatchScheduler({name}=...): a schedule variant Goal Force never uses "
                                          "(FM:41-58); only the shifted linear ramp is built")
        self.num_train_timesteps = num_train_timesteps
        self.shift, self.sigma_max, self.sigma_min = shift, sigma_max, sigma_min
        self.extra_one_step = extra_one_step
        self.training = False
        self.set_timesteps(num_inference_steps)

    # ------------------------------------------------------------------ tables
    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False, shift=None,
                      dynamic_shift_len=None, exponential_shift_mu=None):
        """sigma_i = w(l_i) with l = linspace(start, sigma_min) (one extra point, last dropped, when
        `extra_one_step`) and w(l) = shift*l / (1 + (shift-1)*l); t_i = sigma_i * num_train_timesteps (FM:34-60).
        Training mode adds the per-timestep loss weights (FM:61-67)."""
        if dynamic_shift_len is not None or exponential_shift_mu is not None:
            raise NotImplementedError("dynamic / exponential shift belongs to the unbuilt schedule variants")
        if shift is not None:
            self.shift = shift
        start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        if self.extra_one_step:
            ramp = torch.linspace(start, self.sigma_min, num_inference_steps + 1)[:-1]
        else:
            ramp = torch.linspace(start, self.sigma_min, num_inference_steps)
        self.sigmas = self.shift * ramp / (1 + (self.shift - 1) * ramp)
        self.timesteps = self.sigmas * self.num_train_timesteps
        self.training = bool(training)
        if training:
            # a gaussian bump over the timestep axis centred at n/2, floored at its minimum, normalised to mean 1
            bump = torch.exp(-2 * ((self.timesteps - num_inference_steps / 2) / num_inference_steps) ** 2)
            bump = bump - bump.min()
            self.linear_timesteps_weights = bump * (num_inference_steps / bump.sum())

    def _index(self, timestep) -> int:
        """Position of the table entry nearest to `timestep` (FM:73-75)."""
        t = timestep.detach().to("cpu") if isinstance(timestep, torch.Tensor) else timestep
        return int(torch.argmin((self.timesteps - t).abs()))

    def sigma_pair(self, timestep, to_final=False):
        """(sigma, sigma_next): the step from `timestep` ends at the next table entry, or at 0 after the last (FM:76-80)."""
        i = self._index(timestep)
        last = to_final or i + 1 >= len(self.timesteps)
        return self.sigmas[i], (torch.tensor(0.0) if last else self.sigmas[i + 1])

    # ------------------------------------------------------------------ sampling
    def step(self, model_output, timestep, sample, to_final=False, **kwargs):
        """Euler update prev = sample + model_output * (sigma_next - sigma) (FM:72-82)."""
        sigma, sigma_next = self.sigma_pair(timestep, to_final)
        if sample.is_cuda and sample.dtype == torch.bfloat16:
            out = sample.clone()
            ops.cfg_euler_step(out, model_output.contiguous(), None, 1.0, float(sigma_next - sigma))
            return out
        return sample + model_output * (sigma_next - sigma)      # host tensors (tests, fp32 bookkeeping)

    def return_to_timestep(self, timestep, sample, sample_stablized):
        """FM:85-91 — the model output that would take `sample` to `sample_stablized` from `timestep`: (sample - stabilised) / sigma."""
        sigma = self.sigmas[self._index(timestep)]
        return (sample - sample_stablized) / sigma

    def calculate_shift(self, image_seq_len, base_seq_len: int = 256, max_seq_len: int = 8192, base_shift: float = 0.5,
                        max_shift: float = 0.9):
        """FM:114-125 — the linear sequence-length -> shift map of the `exponential_shift` mode (a mode Goal Force never switches on)."""
        m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
        b = base_shift - m * base_seq_len
        return image_seq_len * m + b

    # ------------------------------------------------------------------ training (GF:180-193)
    def add_noise(self, original_samples, noise, timestep):
        """x_t = (1 - sigma) x_0 + sigma * noise (FM:94-100)."""
        sigma = self.sigmas[self._index(timestep)]
        return (1 - sigma) * original_samples + sigma * noise

    def training_target(self, sample, noise, timestep):
        """velocity target noise - x_0 (FM:103-105)."""
        return noise - sample

    def training_weight(self, timestep):
        """FM:108-111."""
        return self.linear_timesteps_weights[self._index(timestep)]
