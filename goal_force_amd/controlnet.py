"""Goal-Force ControlNet — parameter containers with the reference's names / checkpoint layout
(src/goal_force/wan_video_new.py:49-117; SURVEY.md §5: `controlnet_patch_embedding.patch_embedding.*`,
`controlnet_dit.blocks.{i}.*`, `controlnet_zero_convs_after.{i}.{weight[D,D,1],bias}`).

The ControlNet never sees the noisy latent: control latents -> Conv3d(16->D,(1,2,2)) -> N DiT blocks ->
per-layer zero-init Conv1d(D,D,1), added to x after DiT block i (GF:1489-1522, 1559-1570).  Here the
zero-conv + residual add is ONE GEMM epilogue (x = x + state_i Wz^T + bz).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import GoalForceError
from .dit import DiTBlock, WanModel, param_key


def zero_module(module):
    """GF:40-46."""
    for p in module.parameters():
        p.detach().zero_()
    return module


class ControlNet_DiT(nn.Module):
    """GF:49-69."""

    def __init__(self, num_layers, dim=5120, num_heads=40, ffn_dim=13824, eps=1e-6):
        super().__init__()
        self.num_layers = num_layers
        self.blocks = nn.ModuleList([DiTBlock(False, dim, num_heads, ffn_dim, eps) for _ in range(num_layers)])

    def forward(self, x):
        raise NotImplementedError  # as in the reference (GF:67-68)


class ControlNet_PatchEmbedding(nn.Module):
    """GF:72-94 — (bs,16,f,H,W) -> (bs, f*(H/2)*(W/2), dim)."""

    def __init__(self, in_channels=16, dim=5120, patch_size=(1, 2, 2)):
        super().__init__()
        self.patch_embedding = nn.Conv3d(in_channels, dim, kernel_size=patch_size, stride=patch_size)
        self._patch_w, self._patch_w_key = None, None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 4:
            x = x.unsqueeze(0)
        if x.dim() != 5 or x.shape[0] != 1:
            raise GoalForceError("ControlNet_PatchEmbedding takes [1,16,f,H,W]")
        if self._patch_w is None or self._patch_w_key != param_key(self.patch_embedding.weight):
            self._patch_w = WanModel.padded_patch_weight(self.patch_embedding)      # rebuilt after every weight update
            self._patch_w_key = param_key(self.patch_embedding.weight)
        cols = ops.patchify_im2col(x[0].contiguous(), None, kpad=self._patch_w.shape[1])
        return ops.gemm(cols, self._patch_w, self.patch_embedding.bias).unsqueeze(0)

class ControlNet(nn.Module):
    """GF:97-117 — all trainable ControlNet parameters."""

    def __init__(self, num_layers, stride=None, torch_dtype=torch.bfloat16, dim=5120, num_heads=40, ffn_dim=13824,
                 eps=1e-6):
        super().__init__()
        self.num_layers = num_layers
        self.stride = stride
        self.controlnet_patch_embedding = ControlNet_PatchEmbedding(dim=dim).to(torch_dtype)
        self.controlnet_dit = ControlNet_DiT(num_layers, dim, num_heads, ffn_dim, eps)
        self.controlnet_zero_convs_after = nn.ModuleList(
            [zero_module(nn.Conv1d(dim, dim, kernel_size=1, dtype=torch_dtype)) for _ in range(num_layers)])
        self._is_zero, self._is_zero_key = None, None

    def zero_conv_weight(self, i):
        w = self.controlnet_zero_convs_after[i].weight
        return w.view(w.shape[0], w.shape[1])  # Conv1d k=1 == Linear(D, D)

    def all_zero(self) -> bool:
        """True when every zero-conv is exactly zero (the never-trained low-noise ControlNet2, GF:565,
        INF:108-109): its contribution is x + 0 == x bitwise, so model_fn may skip the ControlNet."""
        key = param_key(*(t for c in self.controlnet_zero_convs_after for t in (c.weight, c.bias)))
        if self._is_zero is None or self._is_zero_key != key:        # re-examined after any update of a zero-conv
            self._is_zero_key = key
            z = True
            for c in self.controlnet_zero_convs_after:
                if bool(c.weight.detach().ne(0).any()) or bool(c.bias.detach().ne(0).any()):
                    z = False
                    break
            self._is_zero = z
        return self._is_zero

    def invalidate(self):
        """Drop everything derived from the parameters (not needed after optimiser steps or load_state_dict — the caches
        compare parameter versions — but harmless, and the explicit hook for code that writes weights behind torch's back)."""
        self._is_zero = None
        self.controlnet_patch_embedding._patch_w = None
