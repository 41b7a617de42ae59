"""Goal-Force control-signal ("force map") videos — host mirror of ControlSignalDataset_Balls
(src/goal_force/unified_dataset.py:616-1080) with the pixel work on the GPU (gf_force_map).

CSV schema (README.md:92-107, DS:942-1026): image, projectile_force_angle, projectile_force_magnitude,
projectile_coordx/y, projectile_mass, target_indirect_force_angle/magnitude, target_coordx/y, target_mass,
width, height, caption.  Channel 0 = direct force blob, 1 = goal (indirect) force blob, 2 = mass blobs.
Output: bf16 [frames, H, W, 3] in [0,1], fed to the VAE un-rescaled (GF:800).

Host side (this file) does exactly the reference's float64 scalar arithmetic for the blob centres /
radii (DS:809-823, 891-901) and rounds each scalar once to fp32, as torch does when a Python float meets
an fp32 tensor; the kernel then evaluates exp(-((x-cx)^2+(y-cy)^2)/denom) per pixel.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ops
from ._lib import GoalForceError

BLOB_RADIUS = 20          # DS:823, 838
MIN_MASS_RADIUS, MAX_MASS_RADIUS = 5, 40   # DS:894-895


class BlobPlan:
    """Flat description of every gaussian of one control video."""

    def __init__(self, frames: int, height: int, width: int):
        self.frames, self.height, self.width = frames, height, width
        self.channels: List[int] = []
        self.params: List[Tuple[float, float]] = []     # (denom = 2 r^2, amplitude)
        self.centers: List[np.ndarray] = []             # [frames, 2] float64 (cx, cy)
        self.clamp01 = False

    def add(self, channel, centers64, radius, amplitude=1.0):
        self.channels.append(int(channel))
        self.params.append((2.0 * radius ** 2, amplitude))
        self.centers.append(np.asarray(centers64, dtype=np.float64))

    def arrays(self):
        n = len(self.channels)
        ch = np.asarray(self.channels, dtype=np.int32)
        pr = np.asarray(self.params, dtype=np.float64).reshape(n, 2).astype(np.float32)
        ce = (np.stack(self.centers, 0) if n else np.zeros((0, self.frames, 2))).astype(np.float32)
        return ch, pr, ce


def plan_control_video(force, angle, x_pos, y_pos, target_indirect_force, target_indirect_angle, target_x_pos,
                       target_y_pos, num_frames, height, width, masses, coords, min_force, max_force,
                       min_indirect_force, max_indirect_force, min_mass, max_mass, p_mask_out_direct_force=0.0,
                       p_mask_out_indirect_force=0.0, p_mask_out_masses=0.0) -> BlobPlan:
    """DS:775-889 — which blobs exist and where (all scalar math in Python floats = float64)."""
    plan = BlobPlan(num_frames, height, width)
    # STEP 1 (DS:785-802): which force channel is masked out
    if force == -1:
        mask_direct, mask_indirect = True, False
    elif target_indirect_force == -1:
        mask_direct, mask_indirect = False, True
    else:
        mask_direct = mask_indirect = False
        u = np.random.uniform(low=0.0, high=1.0)
        if u < p_mask_out_direct_force:
            mask_direct = True
        elif p_mask_out_direct_force <= u <= p_mask_out_direct_force + p_mask_out_indirect_force:
            mask_indirect = True
    disp_max, disp_min = width / 2, width / 8      # DS:804-805

    def moving_blob(channel, xp, yp, f, fmin, fmax, ang):
        x0 = xp * width
        y0 = (1 - yp) * height
        pct = (f - fmin) / (fmax - fmin)
        disp = disp_min + (disp_max - disp_min) * pct
        x1 = x0 + disp * math.cos(ang * torch.pi / 180.0)
        y1 = y0 - disp * math.sin(ang * torch.pi / 180.0)
        cs = np.empty((num_frames, 2), dtype=np.float64)
        for fr in range(num_frames):
            t = fr / (num_frames - 1)
            cs[fr, 0] = x0 * (1 - t) + x1 * t
            cs[fr, 1] = y0 * (1 - t) + y1 * t
        plan.add(channel, cs, BLOB_RADIUS)

    if not mask_direct:       # STEP 2 (DS:808-824)
        moving_blob(0, x_pos, y_pos, force, min_force, max_force, angle)
    if not mask_indirect:     # STEP 3 (DS:827-841)
        moving_blob(1, target_x_pos, target_y_pos, target_indirect_force, min_indirect_force, max_indirect_force,
                    target_indirect_angle)
    # STEP 5 (DS:846-887): static mass blobs in channel 2, then clamp everything to [0,1]
    if not (np.random.uniform(low=0.0, high=1.0) < p_mask_out_masses):
        def mass_blob(x, y, m):
            t = (m - min_mass) / (max_mass - min_mass)
            r = (1 - t) * MIN_MASS_RADIUS + t * MAX_MASS_RADIUS
            plan.add(2, np.tile(np.array([[float(x), float(y)]]), (num_frames, 1)), r)

        if masses["projectile"] > -1:
            mass_blob(coords["projectile"][0], height - coords["projectile"][1], masses["projectile"])
        if masses["target"] > -1:
            mass_blob(coords["target"][0], height - coords["target"][1], masses["target"])
        for m, (xd, yd) in zip(masses["distractors"], coords["distractors"]):
            if m == -1:
                continue
            mass_blob(xd, height - yd, m)
        plan.clamp01 = True
    return plan


def render_control_video(plan: BlobPlan, device="cuda") -> torch.Tensor:
    """BlobPlan -> bf16 [frames,H,W,3] on the GPU (gf_force_map kernel)."""
    ch, pr, ce = plan.arrays()
    return ops.force_map(plan.frames, plan.height, plan.width, torch.from_numpy(ch).to(device),
                         torch.from_numpy(pr).to(device), torch.from_numpy(ce).to(device), plan.clamp01, device)


class ControlSignalDataset_Balls(torch.utils.data.Dataset):
    """Inference-mode (is_validation_dataset=True) mirror of DS:616-1080: one CSV row + its PNG ->
    {"video": [PIL], "prompt", "control_video" bf16 [F,H,W,3], force/angle/... , "masses", "coords"}."""

    def __init__(self, base_path=None, metadata_path=None, repeat=1, data_file_keys=tuple(),
                 main_data_operator=lambda x: x, special_operator_map=None, is_validation_dataset=False,
                 num_frames=None, height=None, width=None, p_mask_out_direct_force=0.0,
                 p_mask_out_indirect_force=0.0, p_mask_out_masses=0.0, device="cuda"):
        if not is_validation_dataset:
            raise NotImplementedError("training-mode (video) datasets are outside the sampling path (SURVEY §2 #7)")
        assert p_mask_out_direct_force + p_mask_out_indirect_force <= 1   # DS:656
        assert 0.0 <= p_mask_out_masses <= 1.0
        self.base_path, self.metadata_path, self.repeat = base_path, metadata_path, repeat
        self.is_validation_dataset = True
        self.num_frames, self.height, self.width = num_frames, height, width
        self.p_mask_out_direct_force = p_mask_out_direct_force
        self.p_mask_out_indirect_force = p_mask_out_indirect_force
        self.p_mask_out_masses = p_mask_out_masses
        self.media_type = "image"
        self.device = device
        self.load_metadata()

    def load_metadata(self):
        """DS:753-773 (validation branch): keep the CSV rows whose image exists; force range 0..1 until the
        driver overwrites it (INF:137-146)."""
        import pandas
        img_dir = os.path.join(self.base_path, "images")
        names = set(os.listdir(img_dir)) if os.path.isdir(img_dir) else set()
        df = pandas.read_csv(self.metadata_path)
        self.df = df[df[self.media_type].map(lambda x: x in names)]
        self.min_force, self.max_force = 0.0, 1.0

    def __len__(self):
        return len(self.df) * self.repeat

    def get_batch(self, idx):
        """DS:942-1026 (image branch)."""
        from PIL import Image
        item = self.df.iloc[idx]
        image = Image.open(os.path.join(self.base_path, "images", item[self.media_type]))
        if image.size != (self.width, self.height):
            image = image.resize((self.width, self.height), resample=Image.Resampling.LANCZOS)
        masses = {"projectile": item["projectile_mass"], "target": item["target_mass"], "distractors": []}
        coords = {"projectile": [int(item["projectile_coordx"]), int(item["projectile_coordy"])],
                  "target": [int(item["target_coordx"]), int(item["target_coordy"])], "distractors": []}
        return dict(image=image, caption=item["caption"], force=item["projectile_force_magnitude"],
                    angle=item["projectile_force_angle"], x_pos=item["projectile_coordx"] / item["width"],
                    y_pos=item["projectile_coordy"] / item["height"],
                    target_indirect_force=item["target_indirect_force_magnitude"],
                    target_indirect_angle=item["target_indirect_force_angle"],
                    target_x_pos=item["target_coordx"] / item["width"], target_y_pos=item["target_coordy"] / item["height"],
                    file_id=str(item[self.media_type]).split(".png")[0], masses=masses, coords=coords)

    def plan(self, b) -> BlobPlan:
        for a in ("min_indirect_force", "max_indirect_force", "min_mass", "max_mass"):
            if not hasattr(self, a):
                raise GoalForceError(f"dataset.{a} must be set by the driver (INF:137-146)")
        return plan_control_video(b["force"], b["angle"], b["x_pos"], b["y_pos"], b["target_indirect_force"],
                                  b["target_indirect_angle"], b["target_x_pos"], b["target_y_pos"], self.num_frames,
                                  self.height, self.width, b["masses"], b["coords"], self.min_force, self.max_force,
                                  self.min_indirect_force, self.max_indirect_force, self.min_mass, self.max_mass,
                                  self.p_mask_out_direct_force, self.p_mask_out_indirect_force, self.p_mask_out_masses)

    def _generate_control_video(self, force, angle, x_pos, y_pos, target_indirect_force, target_indirect_angle,
                                target_x_pos, target_y_pos, num_frames=49, num_channels=3, height=480, width=720,
                                masses={}, coords={}):
        """DS:775-889 signature; returns bf16 [num_frames, height, width, 3] on self.device."""
        plan = plan_control_video(force, angle, x_pos, y_pos, target_indirect_force, target_indirect_angle,
                                  target_x_pos, target_y_pos, num_frames, height, width, masses, coords,
                                  self.min_force, self.max_force, self.min_indirect_force, self.max_indirect_force,
                                  self.min_mass, self.max_mass, self.p_mask_out_direct_force,
                                  self.p_mask_out_indirect_force, self.p_mask_out_masses)
        return render_control_video(plan, self.device)

    def __getitem__(self, data_id):
        b = self.get_batch(data_id % len(self.df))
        control_video = render_control_video(self.plan(b), self.device)
        return {"video": [b["image"]], "prompt": b["caption"], "control_video": control_video, "force": b["force"],
                "angle": b["angle"], "x_pos": b["x_pos"], "y_pos": b["y_pos"],
                "target_indirect_force": b["target_indirect_force"], "target_indirect_angle": b["target_indirect_angle"],
                "target_x_pos": b["target_x_pos"], "target_y_pos": b["target_y_pos"], "file_id": b["file_id"],
                "masses": b["masses"], "coords": b["coords"]}


class ControlSignalDataset_Dominos(ControlSignalDataset_Balls):
    """DS:1099-1555.  The reference's dominos dataset is its balls dataset line for line: the same constructor, CSV columns,
    `get_batch`, `_generate_control_video` (DS:1253-1367 == DS:775-889), `get_blob_for_mass` and `get_gaussian_blob` — only the
    training-mode video slicing differs (DS:1462-1464), which is outside the sampling path like every training-mode (video)
    loader.  `scripts/train/train.py:167-181` muxes it with the balls and plants sets; here it is the same renderer under the
    reference's name (pinned separately: tests/golden/g15 runs the reference's Dominos class itself)."""


class ControlSignalDataset_Plants(torch.utils.data.Dataset):
    """Inference-mode (is_validation_dataset=True) mirror of DS:1557-1894: the plants scenes carry ONE poke — CSV columns image,
    force, angle, coordx, coordy, width, height, caption (DS:1739-1760) — rendered as the moving direct-force blob in channel 0;
    channels 1 and 2 (goal force, masses) stay zero and nothing is clamped (DS:1667-1698).  `min_force` / `max_force` are 0 / 1 until
    the caller sets them (DS:1663-1664), as with the balls set."""

    def __init__(self, base_path=None, metadata_path=None, repeat=1, data_file_keys=tuple(), main_data_operator=lambda x: x,
                 special_operator_map=None, is_validation_dataset=False, num_frames=None, height=None, width=None, device="cuda"):
        if not is_validation_dataset:
            raise NotImplementedError("training-mode (video) datasets are outside the sampling path (SURVEY §2 #7)")
        self.base_path, self.metadata_path, self.repeat = base_path, metadata_path, repeat
        self.is_validation_dataset = True
        self.num_frames, self.height, self.width = num_frames, height, width
        self.media_type = "image"
        self.device = device
        self.load_metadata()

    def load_metadata(self):
        """DS:1638-1664 (validation branch)."""
        import pandas
        img_dir = os.path.join(self.base_path, "images")
        names = set(os.listdir(img_dir)) if os.path.isdir(img_dir) else set()
        df = pandas.read_csv(self.metadata_path)
        self.df = df[df[self.media_type].map(lambda x: x in names)]
        self.min_force, self.max_force = 0.0, 1.0

    def __len__(self):
        return len(self.df) * self.repeat

    def get_batch(self, idx):
        """DS:1739-1760 (image branch)."""
        from PIL import Image
        item = self.df.iloc[idx]
        image = Image.open(os.path.join(self.base_path, "images", item[self.media_type]))
        if image.size != (self.width, self.height):
            image = image.resize((self.width, self.height), resample=Image.Resampling.LANCZOS)
        return dict(image=image, caption=item["caption"], force=item["force"], angle=item["angle"],
                    x_pos=item["coordx"] / item["width"], y_pos=item["coordy"] / item["height"],
                    file_id=str(item[self.media_type]).split(".png")[0])

    def plan(self, force, angle, x_pos, y_pos, num_frames, height, width) -> BlobPlan:
        """DS:1667-1698: one moving blob of radius 20 in channel 0 — the direct-force geometry of DS:1671-1690, which is that of
        the balls set (plan_control_video's STEP 2) with nothing else drawn and no clamp."""
        if force == -1:
            raise GoalForceError("the plants set has no goal-force mode: `force` must be given (DS:1677)")
        plan = plan_control_video(force, angle, x_pos, y_pos, -1, -1, 0.0, 0.0, num_frames, height, width,
                                  {"projectile": -1, "target": -1, "distractors": []}, {"projectile": [0, 0], "target": [0, 0], "distractors": []},
                                  self.min_force, self.max_force, 0.0, 1.0, 0.0, 1.0)
        plan.clamp01 = False             # (a single blob of amplitude 1 never exceeds 1: the clamp would be a no-op; kept off as DS:1696)
        return plan

    def _generate_control_video(self, force, angle, x_pos, y_pos, num_frames=49, num_channels=3, height=480, width=720):
        """DS:1667 signature; returns bf16 [num_frames, height, width, 3] on self.device."""
        if num_channels != 3:
            raise GoalForceError("control videos have 3 channels")
        return render_control_video(self.plan(force, angle, x_pos, y_pos, num_frames, height, width), self.device)

    def __getitem__(self, data_id):
        b = self.get_batch(data_id % len(self.df))
        cv = self._generate_control_video(b["force"], b["angle"], b["x_pos"], b["y_pos"], num_frames=self.num_frames, num_channels=3,
                                          height=self.height, width=self.width)
        return {"video": [b["image"]], "prompt": b["caption"], "control_video": cv, "force": b["force"], "angle": b["angle"],
                "x_pos": b["x_pos"], "y_pos": b["y_pos"], "file_id": b["file_id"]}
