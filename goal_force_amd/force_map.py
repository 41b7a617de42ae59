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
        if fmax == fmin:
            # the reference divides by zero here (DS:812) and renders a video of NaNs, which then trains silently: refused by name instead
            raise GoalForceError(f"control video: the force range is empty (min = max = {fmin}): a training set needs two different force "
                                 "magnitudes in its CSV, an inference driver sets the range itself (INF:137-146)")
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
            if max_mass == min_mass:
                raise GoalForceError(f"control video: the mass range is empty (min = max = {min_mass}); see the force range (DS:893)")
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


def reference_pixel_roundtrip(frames):
    """What the reference's datasets do to every frame they hand over (DS:955-960 / 981-983, 1057-1064): torchvision `ToTensor`
    (uint8 -> fp32 / 255), `2 x - 1`, later `(x + 1) / 2`, torchvision `ToPILImage` (`x.mul(255).byte()`: a TRUNCATING cast).  In
    fp32 the chain is not the identity: 2x - 1 loses the low bits of small x, and the truncation then turns the levels 1 .. 63
    into 0 .. 62 (every other level survives).  So the image the reference conditions on — `data["video"][0]`, INF:153-197 — is
    the file with its dark pixels one level darker; the same arithmetic is run here, in fp32 torch ops, not a table.
    torchvision is not installed in this image: the two transforms are restated from its published functional code
    (`to_tensor`: `img.to(float32).div(255)`; `to_pil_image`: `pic.mul(255).byte()` for float input) — unpinned by a reference run.
    frames: list of PIL images -> list of PIL images."""
    from PIL import Image
    out = []
    for f in frames:
        t = torch.from_numpy(np.array(f.convert("RGB"), dtype=np.uint8)).to(torch.float32).div(255)
        t = 2 * t - 1
        t = (t + 1) / 2
        out.append(Image.fromarray(t.mul(255).byte().numpy()))
    return out


def load_video_frames(path: str):
    """`load_video_to_pil` (DS:16-62: every frame of a clip as RGB PIL images, through cv2.VideoCapture).  Containers readable in
    this image: a DIRECTORY of frame images (sorted by name) and `.npy` / `.npz` holding uint8 [T,H,W,3]; video files go through cv2
    exactly as the reference when it is importable, else through imageio, else they are refused by name (decoding is host I/O)."""
    from PIL import Image
    if os.path.isdir(path):
        names = sorted(f for f in os.listdir(path) if f.rsplit(".", 1)[-1].lower() in ("png", "jpg", "jpeg", "webp"))
        return [Image.open(os.path.join(path, f)).convert("RGB") for f in names]
    if path.endswith((".npy", ".npz")):
        arr = np.load(path)
        arr = arr[arr.files[0]] if hasattr(arr, "files") else arr
        return [Image.fromarray(np.ascontiguousarray(a)) for a in arr]
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        cap = cv2.VideoCapture(path)
        if not cap.isOpened():
            raise IOError(f"Error: Could not open video file at {path}")
        frames = []
        while True:
            ok, frame = cap.read()
            if not ok:
                break
            frames.append(Image.fromarray(cv2.cvtColor(frame, cv2.COLOR_BGR2RGB)))
        cap.release()
        return frames
    try:
        import imageio
    except ImportError as e:
        raise GoalForceError(f"{path}: decoding a video file needs cv2 (as the reference's load_video_to_pil, DS:32) or imageio, neither "
                             "of which is installed here: pass a directory of frame images or a .npy / .npz of uint8 [T,H,W,3]") from e
    reader = imageio.get_reader(path)
    frames = [Image.fromarray(f) for f in reader]
    reader.close()
    return frames


class ControlSignalDataset_Balls(torch.utils.data.Dataset):
    """Mirror of DS:616-1080.  Inference mode (is_validation_dataset=True): one CSV row + its PNG -> {"video": [PIL], "prompt",
    "control_video" bf16 [F,H,W,3], force / angle / ..., "masses", "coords"}.  Training mode (False, as scripts/train/train.py:151-164
    builds it): rows whose `video` file exists under base_path, force / mass ranges read from the CSV (DS:751-760), the clip's frames
    `[::2][-num_frames:]` (DS:985), the control video with the training-time channel masking.  `video_loader(path) -> list of PIL
    frames` replaces `load_video_to_pil` (default: load_video_frames above)."""
    VIDEO_EXT = ".mp4"

    def __init__(self, base_path=None, metadata_path=None, repeat=1, data_file_keys=tuple(),
                 main_data_operator=lambda x: x, special_operator_map=None, is_validation_dataset=False,
                 num_frames=None, height=None, width=None, p_mask_out_direct_force=0.0,
                 p_mask_out_indirect_force=0.0, p_mask_out_masses=0.0, device="cuda", video_loader=None):
        assert p_mask_out_direct_force + p_mask_out_indirect_force <= 1   # DS:656
        assert 0.0 <= p_mask_out_masses <= 1.0
        self.base_path, self.metadata_path, self.repeat = base_path, metadata_path, repeat
        self.is_validation_dataset = bool(is_validation_dataset)
        self.num_frames, self.height, self.width = num_frames, height, width
        self.p_mask_out_direct_force = p_mask_out_direct_force
        self.p_mask_out_indirect_force = p_mask_out_indirect_force
        self.p_mask_out_masses = p_mask_out_masses
        self.media_type = "image" if self.is_validation_dataset else "video"      # DS:661-666
        self.device = device
        self.video_loader = video_loader or load_video_frames
        self.load_metadata()

    def load_metadata(self):
        """DS:739-773.  Validation: keep the CSV rows whose image exists; force range 0..1 until the driver overwrites it
        (INF:137-146).  Training: keep the rows whose clip exists under base_path; the force / goal-force / mass ranges are the
        CSV's own minima and maxima (goal force over the rows that have one)."""
        import pandas
        df = pandas.read_csv(self.metadata_path)
        if self.is_validation_dataset:
            img_dir = os.path.join(self.base_path, "images")
            names = set(os.listdir(img_dir)) if os.path.isdir(img_dir) else set()
            self.df = df[df[self.media_type].map(lambda x: x in names)]
            self.min_force, self.max_force = 0.0, 1.0
            return
        names = set(f for f in os.listdir(self.base_path) if f.endswith(self.VIDEO_EXT)) if os.path.isdir(self.base_path) else set()
        self.df = df[df[self.media_type].map(lambda x: x in names)]
        self.min_force = float(self.df["projectile_force_magnitude"].min())
        self.max_force = float(self.df["projectile_force_magnitude"].max())
        indirect = self.df[self.df["target_indirect_force_magnitude"] > -1]
        self.min_indirect_force = float(indirect["target_indirect_force_magnitude"].min())
        self.max_indirect_force = float(indirect["target_indirect_force_magnitude"].max())
        self.min_mass, self.max_mass = float(self.df["projectile_mass"].min()), float(self.df["projectile_mass"].max())
        self.length = self.df.shape[0]

    def select_frames(self, frames):
        """Which frames of the decoded clip are trained on (DS:985): every second one, the last num_frames of those."""
        return frames[::2][-self.num_frames:]

    def __len__(self):
        return len(self.df) * self.repeat

    def get_batch(self, idx):
        """DS:942-1026."""
        from PIL import Image
        item = self.df.iloc[idx]
        if self.is_validation_dataset:
            image = Image.open(os.path.join(self.base_path, "images", item[self.media_type]))
            if image.size != (self.width, self.height):
                image = image.resize((self.width, self.height), resample=Image.Resampling.LANCZOS)
            frames, ext = [image], ".png"
        else:
            frames, ext = self.select_frames(self.video_loader(os.path.join(self.base_path, item[self.media_type]))), ".mp4"
            image = frames
        masses = {"projectile": item["projectile_mass"], "target": item["target_mass"], "distractors": []}
        coords = {"projectile": [int(item["projectile_coordx"]), int(item["projectile_coordy"])],
                  "target": [int(item["target_coordx"]), int(item["target_coordy"])], "distractors": []}
        return dict(image=image, caption=item["caption"], force=item["projectile_force_magnitude"],
                    angle=item["projectile_force_angle"], x_pos=item["projectile_coordx"] / item["width"],
                    y_pos=item["projectile_coordy"] / item["height"],
                    target_indirect_force=item["target_indirect_force_magnitude"],
                    target_indirect_angle=item["target_indirect_force_angle"],
                    target_x_pos=item["target_coordx"] / item["width"], target_y_pos=item["target_coordy"] / item["height"],
                    file_id=str(item[self.media_type]).split(ext)[0], masses=masses, coords=coords)

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
        frames = b["image"] if isinstance(b["image"], list) else [b["image"]]
        return {"video": reference_pixel_roundtrip(frames), "prompt": b["caption"], "control_video": control_video, "force": b["force"],
                "angle": b["angle"], "x_pos": b["x_pos"], "y_pos": b["y_pos"],
                "target_indirect_force": b["target_indirect_force"], "target_indirect_angle": b["target_indirect_angle"],
                "target_x_pos": b["target_x_pos"], "target_y_pos": b["target_y_pos"], "file_id": b["file_id"],
                "masses": b["masses"], "coords": b["coords"]}


class ControlSignalDataset_Dominos(ControlSignalDataset_Balls):
    """DS:1099-1555.  The reference's dominos dataset is its balls dataset line for line: the same constructor, CSV columns,
    `get_batch`, `_generate_control_video` (DS:1253-1367 == DS:775-889), `get_blob_for_mass` and `get_gaussian_blob` — only the
    training-mode video slicing differs (DS:1462-1464), which is outside the sampling path like every training-mode (video)
    loader.  `scripts/train/train.py:167-181` muxes it with the balls and plants sets; here it is the same renderer under the
    reference's name (pinned separately: tests/golden/g15 runs the reference's Dominos class itself).  Training mode differs in
    the frames taken from the clip: `[14:][0:num_frames]` (DS:1463)."""

    def select_frames(self, frames):
        return frames[14:][0:self.num_frames]


class ControlSignalDataset_Plants(torch.utils.data.Dataset):
    """Mirror of DS:1557-1894: the plants scenes carry ONE poke — CSV columns image / video, force, angle, coordx, coordy, width, height,
    caption (DS:1739-1760) — rendered as the moving direct-force blob in channel 0; channels 1 and 2 (goal force, masses) stay zero
    and nothing is clamped (DS:1667-1698).  Validation mode: `min_force` / `max_force` are 0 / 1 until the caller sets them
    (DS:1663-1664).  Training mode (scripts/train/train.py:182-191): the CSV's own force range, the clip's first num_frames frames,
    and for the `carnation*` clips the random zoom-crop of DS:1773-1834 (crop window containing the poke point, np.random draws in the
    reference's order, bilinear antialiased resize back to the training size, the poke point re-expressed in the crop)."""
    VIDEO_EXT = ".mp4"

    def __init__(self, base_path=None, metadata_path=None, repeat=1, data_file_keys=tuple(), main_data_operator=lambda x: x,
                 special_operator_map=None, is_validation_dataset=False, num_frames=None, height=None, width=None, device="cuda",
                 video_loader=None):
        self.base_path, self.metadata_path, self.repeat = base_path, metadata_path, repeat
        self.is_validation_dataset = bool(is_validation_dataset)
        self.num_frames, self.height, self.width = num_frames, height, width
        self.media_type = "image" if self.is_validation_dataset else "video"
        self.device = device
        self.video_loader = video_loader or load_video_frames
        self.load_metadata()

    def load_metadata(self):
        """DS:1638-1664."""
        import pandas
        df = pandas.read_csv(self.metadata_path)
        if self.is_validation_dataset:
            img_dir = os.path.join(self.base_path, "images")
            names = set(os.listdir(img_dir)) if os.path.isdir(img_dir) else set()
            self.df = df[df[self.media_type].map(lambda x: x in names)]
            self.min_force, self.max_force = 0.0, 1.0
            return
        names = set(f for f in os.listdir(self.base_path) if f.endswith(self.VIDEO_EXT)) if os.path.isdir(self.base_path) else set()
        self.df = df[df[self.media_type].map(lambda x: x in names)]
        self.min_force, self.max_force = float(self.df["force"].min()), float(self.df["force"].max())
        self.length = self.df.shape[0]

    def __len__(self):
        return len(self.df) * self.repeat

    def get_batch(self, idx):
        """DS:1739-1839."""
        from PIL import Image
        item = self.df.iloc[idx]
        out = dict(caption=item["caption"], force=item["force"], angle=item["angle"])
        if self.is_validation_dataset:
            image = Image.open(os.path.join(self.base_path, "images", item[self.media_type]))
            if image.size != (self.width, self.height):
                image = image.resize((self.width, self.height), resample=Image.Resampling.LANCZOS)
            return dict(out, image=image, pixels=None, x_pos=item["coordx"] / item["width"], y_pos=item["coordy"] / item["height"],
                        file_id=str(item[self.media_type]).split(".png")[0])
        frames = self.video_loader(os.path.join(self.base_path, item[self.media_type]))[:self.num_frames]          # DS:1765-1766
        file_id = str(item[self.media_type]).split(".mp4")[0]
        if not file_id.startswith("carnation"):
            return dict(out, image=frames, pixels=None, x_pos=item["coordx"] / item["width"], y_pos=item["coordy"] / item["height"], file_id=file_id)
        # the carnation clips: a random zoom (1.0 .. 1.3) onto a window that keeps the poke point at least 50 px inside (DS:1773-1834)
        px = torch.stack([torch.from_numpy(np.array(f.convert("RGB"), dtype=np.uint8)).permute(2, 0, 1).to(torch.float32).div(255) for f in frames])
        px = 2 * px - 1                                                                                              # DS:1768
        oh, ow = px.shape[-2:]
        coordx, coordy_top = item["coordx"], oh - item["coordy"]
        zoom = np.random.uniform(1.0, 1.3)
        nw, nh = int(ow / zoom), int(oh / zoom)
        max_x, max_y = ow - nw, oh - nh
        min_x, max_x = max(0, int(coordx - nw + 50)), min(max_x, int(coordx - 50))
        min_y, max_y = max(0, int(coordy_top - nh + 50)), min(max_y, int(coordy_top - 50))
        if min_x >= max_x or min_y >= max_y:
            ox, oy = np.random.randint(0, ow - nw + 1), np.random.randint(0, oh - nh + 1)
        else:
            ox, oy = np.random.randint(min_x, max_x + 1), np.random.randint(min_y, max_y + 1)
        px = px[:, :, oy:oy + nh, ox:ox + nw]
        # torchvision.transforms.functional.resize(tensor, [H, W], antialias=True) = bilinear interpolate with antialiasing
        px = torch.nn.functional.interpolate(px, size=[self.height, self.width], mode="bilinear", align_corners=False, antialias=True)
        fx, fy = ((coordx - ox) / nw) * self.width, ((coordy_top - oy) / nh) * self.height
        return dict(out, image=None, pixels=px, x_pos=fx / self.width, y_pos=1.0 - fy / self.height, file_id=file_id)

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
        if b["pixels"] is not None:          # the cropped clip: (x + 1) / 2 and torchvision's ToPILImage (x.mul(255).byte()), DS:1857-1861
            from PIL import Image
            video = [Image.fromarray(((t + 1) / 2).mul(255).byte().permute(1, 2, 0).numpy()) for t in b["pixels"]]
        else:
            video = reference_pixel_roundtrip(b["image"] if isinstance(b["image"], list) else [b["image"]])
        return {"video": video, "prompt": b["caption"], "control_video": cv, "force": b["force"], "angle": b["angle"],
                "x_pos": b["x_pos"], "y_pos": b["y_pos"], "file_id": b["file_id"]}
