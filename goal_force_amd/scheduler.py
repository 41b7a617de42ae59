"""FlowMatchScheduler — drop-in for diffsynth/schedulers/flow_match.py (FM) as used by Goal Force
(`FlowMatchScheduler(shift=5, sigma_min=0.0, extra_one_step=True)`, GF:127; `set_timesteps(50, 1.0,
shift=5.0)`, GF:663; `step`, GF:721).  The sigma/timestep tables are host-side fp32 (bit-exact with the
reference); `step` on GPU tensors runs the gf_cfg_euler_step kernel.
"""
from __future__ import annotations

import math

import torch

from . import ops


class FlowMatchScheduler:
    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False,
                 exponential_shift=False, exponential_shift_mu=None, shift_terminal=None):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.sigma_max = sigma_max
        self.sigma_min = sigma_min
        self.inverse_timesteps = inverse_timesteps
        self.extra_one_step = extra_one_step
        self.reverse_sigmas = reverse_sigmas
        self.exponential_shift = exponential_shift
        self.exponential_shift_mu = exponential_shift_mu
        self.shift_terminal = shift_terminal
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False, shift=None,
                      dynamic_shift_len=None, exponential_shift_mu=None):
        """FM:34-69."""
        if shift is not None:
            self.shift = shift
        sigma_start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        n = num_inference_steps + 1 if self.extra_one_step else num_inference_steps
        sig = torch.linspace(sigma_start, self.sigma_min, n)
        if self.extra_one_step:
            sig = sig[:-1]
        if self.inverse_timesteps:
            sig = torch.flip(sig, dims=[0])
        if self.exponential_shift:
            if exponential_shift_mu is not None:
                mu = exponential_shift_mu
            elif dynamic_shift_len is not None:
                mu = self.calculate_shift(dynamic_shift_len)
            else:
                mu = self.exponential_shift_mu
            sig = math.exp(mu) / (math.exp(mu) + (1 / sig - 1))
        else:
            sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        if self.shift_terminal is not None:
            one_minus_z = 1 - sig
            scale_factor = one_minus_z[-1] / (1 - self.shift_terminal)
            sig = 1 - (one_minus_z / scale_factor)
        if self.reverse_sigmas:
            sig = 1 - sig
        self.sigmas = sig
        self.timesteps = sig * self.num_train_timesteps
        if training:
            x = self.timesteps
            y = torch.exp(-2 * ((x - num_inference_steps / 2) / num_inference_steps) ** 2)
            y_shifted = y - y.min()
            self.linear_timesteps_weights = y_shifted * (num_inference_steps / y_shifted.sum())
            self.training = True
        else:
            self.training = False

    def _timestep_id(self, timestep):
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.cpu()
        return int(torch.argmin((self.timesteps - timestep).abs()))

    def sigma_pair(self, timestep, to_final=False):
        """(sigma, sigma_next) of FM:75-80."""
        tid = self._timestep_id(timestep)
        sigma = self.sigmas[tid]
        if to_final or tid + 1 >= len(self.timesteps):
            sigma_ = torch.tensor(1.0 if (self.inverse_timesteps or self.reverse_sigmas) else 0.0)
        else:
            sigma_ = self.sigmas[tid + 1]
        return sigma, sigma_

    def step(self, model_output, timestep, sample, to_final=False, **kwargs):
        """FM:72-82 — prev = sample + model_output * (sigma_next - sigma)."""
        sigma, sigma_ = self.sigma_pair(timestep, to_final)
        if sample.is_cuda and sample.dtype == torch.bfloat16:
            out = sample.clone()
            ops.cfg_euler_step(out, model_output.contiguous(), None, 1.0, float(sigma_ - sigma))
            return out
        return sample + model_output * (sigma_ - sigma)  # host tensors (tests, fp32 bookkeeping)

    def return_to_timestep(self, timestep, sample, sample_stablized):
        sigma = self.sigmas[self._timestep_id(timestep)]
        return (sample - sample_stablized) / sigma

    def add_noise(self, original_samples, noise, timestep):
        sigma = self.sigmas[self._timestep_id(timestep)]
        return (1 - sigma) * original_samples + sigma * noise

    def training_target(self, sample, noise, timestep):
        return noise - sample

    def training_weight(self, timestep):
        tid = int(torch.argmin((self.timesteps - timestep.to(self.timesteps.device)).abs()))
        return self.linear_timesteps_weights[tid]

    def calculate_shift(self, image_seq_len, base_seq_len: int = 256, max_seq_len: int = 8192,
                        base_shift: float = 0.5, max_shift: float = 0.9):
        m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
        b = base_shift - m * base_seq_len
        return image_seq_len * m + b
