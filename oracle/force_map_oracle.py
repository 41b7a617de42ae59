"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Torch-CPU restatement of the Goal-Force control-signal video
(src/goal_force/unified_dataset.py:775-940: _generate_control_video / get_gaussian_blob /
get_blob_for_mass), pinned bit-exactly (sha256) against tests/golden/g7_force_maps.npz which the
reference's own class produced for all 12 example CSV rows + 2 synthetic direct-force rows.
`plants_control_video` restates the plants variant (DS:1667-1698), pinned the same way against tests/golden/g15_dataset_variants.npz
(the reference's ControlSignalDataset_Plants / _Dominos classes run on seeded rows)."""
import math

import numpy as np
import torch


def gaussian_blob(x, y, radius, height, width):
    """DS:903-940 — exp(-((xg-x)^2+(yg-y)^2)/(2 r^2)) on an int64 grid (fp32 arithmetic, true division)."""
    yg, xg = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    sq = (xg - x) ** 2 + (yg - y) ** 2
    return 1.0 * torch.exp(-sq / (2.0 * radius ** 2))


def control_video(force, angle, x_pos, y_pos, tforce, tangle, tx, ty, masses, coords, num_frames=81, height=480,
                  width=832, min_force=30.0, max_force=400.0, min_mass=1.0, max_mass=4.0, p_mask_out_direct_force=0.0,
                  p_mask_out_indirect_force=0.0, p_mask_out_masses=0.0):
    """DS:775-889.  With the mask-out probabilities at 0 (inference) nothing random happens; with the training script's values the
    two np.random.uniform draws of DS:796-801 and DS:852 decide which channels survive (pinned by g15's masked rows).  Returns bf16 [F,H,W,3]."""
    sig = torch.zeros((num_frames, 3, height, width))
    if force == -1:
        mask_direct, mask_indirect = True, False
    elif tforce == -1:
        mask_direct, mask_indirect = False, True
    else:
        mask_direct = mask_indirect = False
        u = np.random.uniform(low=0.0, high=1.0)                 # drawn whatever the probabilities are (DS:796)
        if u < p_mask_out_direct_force:
            mask_direct = True
        elif p_mask_out_direct_force <= u <= p_mask_out_direct_force + p_mask_out_indirect_force:
            mask_indirect = True

    def moving(ch, xp, yp, f, ang):
        x0, y0 = xp * width, (1 - yp) * height
        disp = width / 8 + (width / 2 - width / 8) * ((f - min_force) / (max_force - min_force))
        x1 = x0 + disp * math.cos(ang * torch.pi / 180.0)
        y1 = y0 - disp * math.sin(ang * torch.pi / 180.0)
        for fr in range(num_frames):
            t = fr / (num_frames - 1)
            sig[fr, ch] += gaussian_blob(x0 * (1 - t) + x1 * t, y0 * (1 - t) + y1 * t, 20, height, width)

    if not mask_direct:
        moving(0, x_pos, y_pos, force, angle)
    if not mask_indirect:
        moving(1, tx, ty, tforce, tangle)
    sig = sig.permute(0, 2, 3, 1).contiguous()
    sig[..., 2] = 0

    def mass(x, y, m):
        t = (m - min_mass) / (max_mass - min_mass)
        return gaussian_blob(x, y, (1 - t) * 5 + t * 40, height, width)[None]

    if np.random.uniform(low=0.0, high=1.0) < p_mask_out_masses:         # drawn whatever the probability is (DS:852)
        return sig.to(torch.bfloat16)                            # masses masked out: no blobs in channel 2 and NO clamp (DS:853, 887)
    if masses["projectile"] > -1:
        sig[..., 2] += mass(coords["projectile"][0], height - coords["projectile"][1], masses["projectile"])
    if masses["target"] > -1:
        sig[..., 2] += mass(coords["target"][0], height - coords["target"][1], masses["target"])
    return torch.clamp(sig, min=0.0, max=1.0).to(torch.bfloat16)


def plants_control_video(force, angle, x_pos, y_pos, num_frames=49, height=480, width=720, min_force=30.0, max_force=400.0):
    """DS:1667-1698: the direct-force blob alone, written to all three channels and then channels 1, 2 zeroed; no clamp.  bf16 [F,H,W,3]."""
    sig = torch.zeros((num_frames, 3, height, width))
    x0, y0 = x_pos * width, (1 - y_pos) * height
    disp = width / 8 + (width / 2 - width / 8) * ((force - min_force) / (max_force - min_force))
    x1 = x0 + disp * math.cos(angle * torch.pi / 180.0)
    y1 = y0 - disp * math.sin(angle * torch.pi / 180.0)
    for fr in range(num_frames):
        t = fr / (num_frames - 1)
        sig[fr] += gaussian_blob(x0 * (1 - t) + x1 * t, y0 * (1 - t) + y1 * t, 20, height, width)[None]
    sig = sig.permute(0, 2, 3, 1).contiguous()
    sig[..., 1:3] = 0
    return sig.to(torch.bfloat16)
