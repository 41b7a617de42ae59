"""Canny-edge control signal — host mirror of `ControlSignalDataset_CannyEdge` (src/goal_force/unified_dataset.py:406-613) and
of the `controlnet_aux.CannyDetector` call it makes per frame, with the pixel work on the GPU (csrc/gf_canny.hip).

What the reference computes for a clip (`_generate_control_video`, DS:559-578), per frame:
    canny = CannyDetector()(frame)          # controlnet_aux defaults: low 100, high 200, detect_resolution = image_resolution = 512
        img  = resize_image(HWC3(frame), 512)   # short side -> 512, both sides rounded to multiples of 64;
                                                #   cv2.INTER_LANCZOS4 when enlarging (k > 1), cv2.INTER_AREA otherwise
        map  = HWC3(cv2.Canny(img, 100, 200))   # aperture 3, L1 gradient; grey -> three equal channels
        map  = cv2.resize(map, (W', H'), INTER_LINEAR) with (H', W') = resize_image(frame, 512).shape: the SAME size -> a copy
    if canny.shape[:2] != frame.shape[:2]: canny = cv2.resize(canny, (W, H), interpolation=cv2.INTER_AREA)
then `stack -> float32 / 127.5 - 1.0 -> bf16` [frames, H, W, 3].

cv2 and controlnet_aux are NOT in this image: the class cannot be imported here and nothing it computes can be pinned against
it.  What is built is OpenCV's published algorithm (imgproc resize.cpp / canny.cpp) restated integer for integer — "parity
unpinned (cv2 absent)"; the HIP kernels are tested bit-exact against an independent numpy restatement (tests/test_canny.py).
Video decoding (`LoadVideo`, imageio) is host I/O outside the path: `main_data_operator` is injectable exactly as in the
reference (default: identity — pass already decoded frames).
"""
from __future__ import annotations

import json
import math
import os
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
from ._lib import GoalForceError

COEF_BITS = 11                     # INTER_RESIZE_COEF_BITS
_PI = 3.1415926535897932384626433832795   # CV_PI


def resized_shape(h: int, w: int, resolution: int = 512):
    """controlnet_aux.util.resize_image's target size: k = resolution / min(H, W); sides = round(side * k / 64) * 64 (np.round:
    half to even).  Returns (H', W', k)."""
    k = float(resolution) / min(h, w)
    return int(np.round(h * k / 64.0)) * 64, int(np.round(w * k / 64.0)) * 64, k


def lanczos4_tables(ssize: int, dsize: int):
    """resize.cpp, INTER_LANCZOS4 branch: per destination index the floor source coordinate and the 8 tap weights (taps at
    s - 3 .. s + 4) as 11-bit fixed point.  fx = (float)((d + 0.5) * scale - 0.5), weights by interpolateLanczos4 (float32 with
    double intermediates, normalised to sum 1), then saturate_cast<short>(w * 2048) (round half to even)."""
    scale = 1.0 / (float(dsize) / float(ssize))
    s45 = 0.70710678118654752440084436210485
    cs = ((1, 0), (-s45, -s45), (0, 1), (s45, -s45), (-1, 0), (s45, s45), (0, -1), (-s45, s45))
    ofs = np.zeros(dsize, np.int32)
    co = np.zeros((dsize, 8), np.int16)
    for d in range(dsize):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        x = np.float32(fx - np.float32(sx))
        y0 = -float(np.float32(x + np.float32(3))) * _PI * 0.25      # (x + 3) is a float32 sum in the source
        s0, c0 = math.sin(y0), math.cos(y0)
        w = np.zeros(8, np.float32)
        for i in range(8):
            y0_ = np.float32(np.float32(x + np.float32(3)) - np.float32(i))
            if abs(float(y0_)) >= 1e-6:
                y = -float(y0_) * _PI * 0.25
                w[i] = np.float32((cs[i][0] * s0 + cs[i][1] * c0) / (y * y))
            else:
                w[i] = np.float32(1e30)
        tot = np.float32(0)
        for i in range(8):
            tot = np.float32(tot + w[i])
        inv = np.float32(np.float32(1.0) / tot)
        w = (w * inv).astype(np.float32)
        ofs[d] = sx
        co[d] = np.clip(np.rint(w * np.float32(1 << COEF_BITS)), -32768, 32767).astype(np.int16)
    return ofs, co


def area_tables(ssize: int, dsize: int):
    """resize.cpp computeResizeAreaTab (shrinking, non-integer ratio): CSR lists of (source index, fp32 weight) per destination
    index — the overlap of the destination cell [d * scale, (d + 1) * scale) with every source cell, over the cell width."""
    scale = float(ssize) / float(dsize)
    start, src, alpha = [0], [], []
    for d in range(dsize):
        fsx1 = d * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            src.append(sx1 - 1)
            alpha.append(np.float32((sx1 - fsx1) / cell))
        for sx in range(sx1, sx2):
            src.append(sx)
            alpha.append(np.float32(1.0 / cell))
        if fsx2 - sx2 > 1e-3:
            src.append(sx2)
            alpha.append(np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell))
        start.append(len(src))
    return np.asarray(start, np.int32), np.asarray(src, np.int32), np.asarray(alpha, np.float32)


def _identity_tables(n: int):
    return np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.ones(n, np.float32)


def _integer_ratio(ssize: int, dsize: int) -> bool:
    ratio = ssize / dsize
    return abs(ratio - round(ratio)) < 2.220446049250313e-16


def _area_or_identity(ssize: int, dsize: int):
    if ssize == dsize:
        return _identity_tables(ssize)         # cv2.resize to the same size is a copy
    if dsize > ssize:
        raise NotImplementedError("INTER_AREA enlarging (OpenCV switches to a bilinear variant) is not on the Canny path")
    return area_tables(ssize, dsize)


def area_tables_2d(W: int, Wd: int, H: int, Hd: int):
    """The four table triples of one INTER_AREA shrink, chosen PER IMAGE as resize.cpp does: OpenCV takes its integer fast path
    (ResizeAreaFast: integer block sums, one rounding division) only when BOTH scale_x and scale_y are integers; a mixed case — an
    integer ratio on one axis, a fractional one on the other — runs the general overlap tables on both axes, which area_tables
    produces for any ratio.  Refused (not restated here): both ratios integral and at least one > 1, i.e. inputs whose short side
    is an integer multiple of the detect resolution AND whose long side divides evenly too (1024 x 1024, 1024 x 2048, ... at 512)."""
    if (W, H) != (Wd, Hd) and Wd <= W and Hd <= H and _integer_ratio(W, Wd) and _integer_ratio(H, Hd):
        raise NotImplementedError(f"INTER_AREA {W}x{H} -> {Wd}x{Hd}: both ratios are integers ({W // Wd}, {H // Hd}) — OpenCV takes its "
                                  "integer fast path (ResizeAreaFast), which is not restated here")
    return (*_area_or_identity(W, Wd), *_area_or_identity(H, Hd))


def _dev(a: np.ndarray, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _u8_frames(frames, device) -> torch.Tensor:
    """list of PIL images / HWC arrays, or an array / tensor [T,H,W,3] (or [T,H,W], [T,H,W,1]: HWC3 replicates grey) -> uint8
    device tensor [T,H,W,3]."""
    if torch.is_tensor(frames):
        t = frames
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.array(frames)))         # DS:563: np.array(processed_video)
    if t.dtype != torch.uint8:
        raise GoalForceError(f"Canny input must be uint8 frames in [0, 255], got {t.dtype}")
    if t.dim() == 3:
        t = t.unsqueeze(-1)
    if t.dim() != 4 or t.shape[-1] not in (1, 3):
        raise GoalForceError(f"Canny input must be [T,H,W,3] (or grey [T,H,W]) uint8, got {tuple(t.shape)} (RGBA is not used by the dataset)")
    if t.shape[-1] == 1:
        t = t.expand(-1, -1, -1, 3)
    return t.contiguous().to(device)


class CannyDetector:
    """`controlnet_aux.CannyDetector` for batches of frames on the GPU.  __call__(frames uint8 [T,H,W,3]) -> uint8 [T,H',W',3]
    (255 on edges) with (H', W') = resized_shape(H, W, detect_resolution)."""

    MAX_HYSTERESIS_PASSES = 4096

    def __init__(self, device="cuda"):
        self.device = torch.device(device)

    def resize(self, frames, detect_resolution=512):
        """controlnet_aux `resize_image`: uint8 [T,H,W,3] -> uint8 [T,H',W',3] (Lanczos-4 when enlarging, area when shrinking)."""
        x = _u8_frames(frames, self.device)
        if not x.is_cuda:
            raise GoalForceError("CannyDetector: no CPU fallback exists — frames must go to a HIP device")
        T, H, W, _ = x.shape
        Hd, Wd, k = resized_shape(H, W, detect_resolution)
        lib = _lib.load()
        st = torch.cuda.current_stream(x.device).cuda_stream
        if (Hd, Wd) == (H, W):
            return x
        img = torch.empty((T, Hd, Wd, 3), dtype=torch.uint8, device=x.device)
        if k > 1:
            xo, xc = lanczos4_tables(W, Wd)
            yo, yc = lanczos4_tables(H, Hd)
            tabs = [_dev(a, x.device) for a in (xo, xc, yo, yc)]
            _lib.check(lib.gf_resize_lanczos4_u8(x.data_ptr(), img.data_ptr(), *(t.data_ptr() for t in tabs), T, H, W, Hd, Wd, st),
                       "gf_resize_lanczos4_u8")
        else:
            tabs = [_dev(a, x.device) for a in area_tables_2d(W, Wd, H, Hd)]
            _lib.check(lib.gf_resize_area_u8(x.data_ptr(), img.data_ptr(), *(t.data_ptr() for t in tabs), T, H, W, Hd, Wd, 0, st),
                       "gf_resize_area_u8")
        return img

    def canny(self, img, low_threshold=100, high_threshold=200):
        """cv2.Canny on a batch: uint8 [T,H,W,3] on the device -> state uint8 [T,H,W] (2 on edge pixels)."""
        img = _u8_frames(img, self.device)
        T, Hd, Wd, _ = img.shape
        st = torch.cuda.current_stream(img.device).cuda_stream
        state = torch.empty((T, Hd, Wd), dtype=torch.uint8, device=img.device)
        ws = torch.empty((2, T * Hd * Wd), dtype=torch.int32, device=img.device)
        flag = torch.zeros((1,), dtype=torch.int32, device=img.device)
        _lib.check(_lib.load().gf_canny_u8(img.data_ptr(), state.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), flag.data_ptr(), T, Hd, Wd,
                                           int(math.floor(low_threshold)), int(math.floor(high_threshold)), self.MAX_HYSTERESIS_PASSES, st),
                   "gf_canny_u8")
        return state

    def edge_state(self, frames, low_threshold=100, high_threshold=200, detect_resolution=512):
        """-> (state uint8 [T,H',W'] with 2 on edge pixels, (H', W'))."""
        img = self.resize(frames, detect_resolution)
        return self.canny(img, low_threshold, high_threshold), (img.shape[1], img.shape[2])

    def __call__(self, frames, low_threshold=100, high_threshold=200, detect_resolution=512, image_resolution=512):
        if image_resolution != detect_resolution:
            raise NotImplementedError("image_resolution != detect_resolution (a bilinear resize of the edge map) is not used by the dataset")
        state, _ = self.edge_state(frames, low_threshold, high_threshold, detect_resolution)
        return ((state == 2).to(torch.uint8) * 255).unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()


class VideoOperator:
    """`ControlSignalDataset_CannyEdge.default_video_operator(...)` (DS:441-461): path (relative to base_path) -> list of PIL frames,
    each through ImageCropAndResize (DS:136-170: scale = max(W'/W, H'/H), bilinear resize to (round(H*scale), round(W*scale)), centre
    crop).  torchvision's PIL path is `img.resize((w, h), BILINEAR)` and a crop at int(round((size - crop) / 2.0)) — restated on PIL
    alone (torchvision is not in this image).  Frame count: LoadVideo.get_num_frames (DS:188-194): `num_frames`, or for a shorter clip
    the largest count <= its length with count % time_division_factor == time_division_remainder.
    Containers: still images (one frame), GIF (PIL), a DIRECTORY of frame images (sorted by name), `.npy` / `.npz` holding uint8
    [T,H,W,3]; mp4 / avi / mov / ... go through `imageio` exactly as DS:199-207 when it is importable and are refused by name when
    it is not (decoding is host I/O outside the path).  A clip that fails to load returns None, as DS:209-212 does."""

    IMAGE_EXT = ("jpg", "jpeg", "png", "webp")
    VIDEO_EXT = ("mp4", "avi", "mov", "wmv", "mkv", "flv", "webm")

    def __init__(self, base_path="", max_pixels=1920 * 1080, height=None, width=None, height_division_factor=16,
                 width_division_factor=16, num_frames=81, time_division_factor=4, time_division_remainder=1):
        self.base_path, self.max_pixels, self.height, self.width = base_path, max_pixels, height, width
        self.hdiv, self.wdiv = height_division_factor, width_division_factor
        self.num_frames, self.tdiv, self.trem = num_frames, time_division_factor, time_division_remainder

    def target_size(self, image):
        """DS:155-165."""
        if self.height is not None and self.width is not None:
            return self.height, self.width
        width, height = image.size
        if width * height > self.max_pixels:
            scale = (width * height / self.max_pixels) ** 0.5
            height, width = int(height / scale), int(width / scale)
        return height // self.hdiv * self.hdiv, width // self.wdiv * self.wdiv

    def crop_and_resize(self, image):
        """DS:144-153."""
        from PIL import Image
        th, tw = self.target_size(image)
        width, height = image.size
        scale = max(tw / width, th / height)
        image = image.resize((round(width * scale), round(height * scale)), Image.BILINEAR)
        w, h = image.size
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return image.crop((left, top, left + tw, top + th))

    def count(self, available):
        """DS:188-194."""
        n = self.num_frames
        if int(available) < n:
            n = int(available)
            while n > 1 and n % self.tdiv != self.trem:
                n -= 1
        return n

    def __call__(self, data: str):
        from PIL import Image
        path = os.path.join(self.base_path, data)                              # ToAbsolutePath (DS:92-97)
        ext = path.rsplit(".", 1)[-1].lower() if "." in os.path.basename(path) else ""
        try:
            if os.path.isdir(path):
                names = sorted(f for f in os.listdir(path) if f.rsplit(".", 1)[-1].lower() in self.IMAGE_EXT)
                frames = [Image.open(os.path.join(path, f)).convert("RGB") for f in names[: self.count(len(names))]]
            elif ext in ("npy", "npz"):
                arr = np.load(path)
                arr = arr[arr.files[0]] if hasattr(arr, "files") else arr
                frames = [Image.fromarray(np.ascontiguousarray(a)) for a in arr[: self.count(len(arr))]]
            elif ext in self.IMAGE_EXT:
                return [self.crop_and_resize(Image.open(path).convert("RGB"))]
            elif ext == "gif":
                gif = Image.open(path)
                frames = []
                for i in range(self.count(getattr(gif, "n_frames", 1))):
                    gif.seek(i)
                    frames.append(gif.convert("RGB"))
            elif ext in self.VIDEO_EXT:
                try:
                    import imageio
                except ImportError as e:
                    raise GoalForceError(f"{path}: decoding .{ext} needs `imageio` (as the reference's LoadVideo, DS:199), which is not "
                                         "installed here: pass a directory of frame images or a .npy / .npz of uint8 [T,H,W,3]") from e
                reader = imageio.get_reader(path)
                frames = [Image.fromarray(reader.get_data(i)) for i in range(self.count(reader.count_frames()))]
                reader.close()
            else:
                raise GoalForceError(f"{path}: not a frame directory, image, gif, video or .npy / .npz clip")
        except GoalForceError:
            raise
        except Exception as e:      # noqa: BLE001 — DS:209-212: log and mark the sample invalid
            print(f"WARNING: Skipping corrupted or unreadable video file: {path}. Error: {e}")
            return None
        return [self.crop_and_resize(f) for f in frames]


class ControlSignalDataset_CannyEdge(torch.utils.data.Dataset):
    """DS:406-613 — same constructor arguments, `process_for_validation(video_path, prompt)`, `_generate_control_video(frames)`,
    `__getitem__` / `__len__`.  Metadata: `.json` / `.jsonl` lists of dicts, or a CSV (the reference's OpenVid-1M listing) filtered
    to the files that exist under `base_path` (the reference additionally pickles that filtered list as a cache; not kept).
    `metadata_path=None` (the reference's pre-computed `.pth` cache mode) lists `.pth` files under base_path and torch.load's them."""

    def __init__(self, base_path=None, metadata_path=None, repeat=1, data_file_keys=tuple(), main_data_operator: Callable = lambda x: x,
                 special_operator_map=None, device="cuda"):
        self.base_path, self.metadata_path, self.repeat = base_path, metadata_path, repeat
        self.data_file_keys = data_file_keys
        self.main_data_operator = main_data_operator
        self.special_operator_map = {} if special_operator_map is None else special_operator_map
        self.data, self.cached_data = [], []
        self.load_from_cache = metadata_path is None
        self.load_metadata(metadata_path)
        self.canny_detector = CannyDetector(device)

    @staticmethod
    def default_video_operator(base_path="", max_pixels=1920 * 1080, height=None, width=None, height_division_factor=16,
                               width_division_factor=16, num_frames=81, time_division_factor=4, time_division_remainder=1):
        """DS:441-461 — the operator the Canny launcher builds per row (scripts/inference/inference_canny_edge_control.py:141-152)."""
        return VideoOperator(base_path, max_pixels, height, width, height_division_factor, width_division_factor, num_frames,
                             time_division_factor, time_division_remainder)

    def load_metadata(self, metadata_path: Optional[str]):
        if metadata_path is None:
            if self.base_path is not None:
                for root, _, files in os.walk(self.base_path):
                    self.cached_data += [os.path.join(root, f) for f in sorted(files) if f.endswith(".pth")]
        elif metadata_path.endswith(".json"):
            with open(metadata_path) as f:
                self.data = json.load(f)
        elif metadata_path.endswith(".jsonl"):
            with open(metadata_path) as f:
                self.data = [json.loads(line.strip()) for line in f if line.strip()]
        else:
            import pandas
            meta = pandas.read_csv(metadata_path)
            self.data = [row for row in (meta.iloc[i].to_dict() for i in range(len(meta)))
                         if os.path.exists(os.path.join(self.base_path or "", str(row["video"])))]      # DS:507-514

    def _generate_control_video(self, processed_video):
        """DS:559-578: frames (list of PIL / uint8 arrays, or [T,H,W,3] uint8) -> bf16 [T,H,W,3] in [-1, 1] on the device."""
        det = self.canny_detector
        x = _u8_frames(processed_video, det.device)
        T, H, W, _ = x.shape
        state, (Hd, Wd) = det.edge_state(x)
        tabs = [_dev(a, x.device) for a in area_tables_2d(Wd, W, Hd, H)]
        out = torch.empty((T, H, W, 3), dtype=torch.bfloat16, device=x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.load().gf_resize_area_u8(state.data_ptr(), out.data_ptr(), *(t.data_ptr() for t in tabs), T, Hd, Wd, H, W, 1, st),
                   "gf_resize_area_u8")
        return out

    def process_for_validation(self, video_path, prompt: str):
        """DS:533-553."""
        return {"prompt": prompt, "control_signal_video": self._generate_control_video(self.main_data_operator(video_path))}

    def __getitem__(self, data_id):
        if self.load_from_cache:
            return torch.load(self.cached_data[data_id % len(self.cached_data)], map_location="cpu", weights_only=False)
        data = dict(self.data[data_id % len(self.data)])
        for key in self.data_file_keys:
            if key in data:
                if key in self.special_operator_map:
                    data[key] = self.special_operator_map[key]         # (sic, DS:590: the operator object itself)
                else:
                    processed = self.main_data_operator(data[key])
                    if processed is None:
                        return None                                    # a clip that failed to load marks the sample invalid
                    data[key] = processed
        data["control_video"] = self._generate_control_video(data["video"])
        data["prompt"] = data.pop("caption")
        return data

    def __len__(self):
        return (len(self.cached_data) if self.load_from_cache else len(self.data)) * self.repeat
