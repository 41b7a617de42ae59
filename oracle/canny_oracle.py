"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of what `ControlSignalDataset_CannyEdge._generate_control_video`
(src/goal_force/unified_dataset.py:559-578) computes per frame through controlnet_aux.CannyDetector and OpenCV.

**PARITY UNPINNED**: cv2 and controlnet_aux are absent from this image, the reference class cannot be imported, and the
reference holds no Canny fixtures.  What is restated is the PUBLISHED algorithm of the third-party code the reference calls:
  * controlnet_aux (unpinned in the reference's requirements) `CannyDetector.__call__` defaults (low 100, high 200, detect /
    image resolution 512) and `util.resize_image` / `HWC3`;
  * OpenCV (`opencv-python`, unpinned) imgproc: `resize` INTER_LANCZOS4 / INTER_AREA for 8-bit images (resize.cpp:
    interpolateLanczos4, 11-bit fixed-point taps, FixedPtCast shift 22; computeResizeAreaTab + ResizeArea_) and `Canny`
    (canny.cpp: Sobel 3x3 with BORDER_REPLICATE, L1 magnitude, strongest channel, NMS with the TG22 fixed-point tangent test,
    thresholds `m > low`, `m > high`, 8-connected hysteresis).
Written independently of goal_force_amd/canny.py (vectorised, different code); tests/test_canny.py pins it with hand-computed
known-answer cases and compares the HIP kernels with it bit for bit.  Nothing under goal_force_amd/ may import this file."""
import math

import numpy as np


def resize_target(h, w, resolution=512):
    k = float(resolution) / min(h, w)
    return int(np.round(h * k / 64.0)) * 64, int(np.round(w * k / 64.0)) * 64, k


# ------------------------------------------------------------------ INTER_LANCZOS4, 8-bit
def _lanczos4_weights(x):
    """interpolateLanczos4 for an array of fractional positions x (float32) -> [n, 8] float32."""
    x = x.astype(np.float32)
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]], np.float64)
    xp3 = (x + np.float32(3)).astype(np.float32)
    y0 = -xp3.astype(np.float64) * math.pi * 0.25
    s0, c0 = np.sin(y0), np.cos(y0)
    w = np.empty((x.size, 8), np.float32)
    for i in range(8):
        y0_ = (xp3 - np.float32(i)).astype(np.float32)
        y = -y0_.astype(np.float64) * math.pi * 0.25
        with np.errstate(divide="ignore", invalid="ignore"):
            v = ((cs[i, 0] * s0 + cs[i, 1] * c0) / (y * y)).astype(np.float32)
        w[:, i] = np.where(np.abs(y0_) >= np.float32(1e-6), v, np.float32(1e30))
    tot = np.zeros(x.size, np.float32)
    for i in range(8):
        tot = (tot + w[:, i]).astype(np.float32)
    return (w * (np.float32(1.0) / tot)[:, None]).astype(np.float32)


def _lanczos_axis(ssize, dsize):
    scale = 1.0 / (float(dsize) / float(ssize))
    fx = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    w = _lanczos4_weights((fx - sx.astype(np.float32)).astype(np.float32))
    co = np.clip(np.rint(w * np.float32(2048)), -32768, 32767).astype(np.int64)
    idx = np.clip(sx[:, None] + np.arange(-3, 5)[None, :], 0, ssize - 1)
    return idx, co


def resize_lanczos4_u8(img, hd, wd):
    """img uint8 [H,W,C] -> uint8 [hd,wd,C]."""
    h, w, _ = img.shape
    xi, xc = _lanczos_axis(w, wd)
    yi, yc = _lanczos_axis(h, hd)
    src = img.astype(np.int64)
    hor = np.zeros((h, wd, img.shape[2]), np.int64)
    for k in range(8):
        hor += src[:, xi[:, k], :] * xc[:, k][None, :, None]
    ver = np.zeros((hd, wd, img.shape[2]), np.int64)
    for k in range(8):
        ver += hor[yi[:, k], :, :] * yc[:, k][:, None, None]
    return np.clip((ver + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ INTER_AREA (shrinking, non-integer ratio), 8-bit
def _area_axis(ssize, dsize):
    """per-row ordered entry lists (source index, fp32 weight) of resize.cpp computeResizeAreaTab (the order matters for the float
    sums).  Valid for any shrink ratio; whether OpenCV USES these tables is decided per image (resize_area_u8): only when both
    ratios are integers does it take ResizeAreaFast instead, which is not restated."""
    if ssize == dsize:
        return [[(i, np.float32(1.0))] for i in range(ssize)]
    scale = float(ssize) / float(dsize)
    assert scale > 1, "only shrinking is on the Canny path"
    rows = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(math.ceil(f1)), int(math.floor(f2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        ent = []
        if s1 - f1 > 1e-3:
            ent.append((s1 - 1, np.float32((s1 - f1) / cell)))
        ent += [(s, np.float32(1.0 / cell)) for s in range(s1, s2)]
        if f2 - s2 > 1e-3:
            ent.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
        rows.append(ent)
    return rows


def resize_area_u8(img, hd, wd):
    """img uint8 [H,W,C] -> uint8 [hd,wd,C]: horizontal float sums per source row, then the vertical combination, cvRound."""
    h, w, c = img.shape
    integral = lambda a, b: abs(a / b - round(a / b)) < 2.220446049250313e-16       # noqa: E731
    assert (h, w) == (hd, wd) or not (integral(w, wd) and integral(h, hd)), "both ratios integral: OpenCV's ResizeAreaFast is not restated"
    xr, yr = _area_axis(w, wd), _area_axis(h, hd)
    src = img.astype(np.float32)
    hor = np.zeros((h, wd, c), np.float32)
    nmax = max(len(e) for e in xr)
    for j in range(nmax):                                       # j-th entry of every destination column, in order
        idx = np.array([e[j][0] if j < len(e) else 0 for e in xr])
        al = np.array([e[j][1] if j < len(e) else np.float32(0) for e in xr], np.float32)
        has = np.array([j < len(e) for e in xr])
        term = (src[:, idx, :] * al[None, :, None]).astype(np.float32)
        hor = np.where(has[None, :, None], (hor + term).astype(np.float32), hor)
    out = np.zeros((hd, wd, c), np.float32)
    for d, ent in enumerate(yr):
        acc = None
        for s, b in ent:
            t = (hor[s] * b).astype(np.float32)
            acc = t if acc is None else (acc + t).astype(np.float32)
        out[d] = acc
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ cv2.Canny (aperture 3, L1 gradient)
def canny_u8(img, low=100, high=200):
    """img uint8 [H,W,C] -> uint8 [H,W] (255 on edges)."""
    from scipy import ndimage
    h, w, c = img.shape
    p = np.pad(img.astype(np.int32), ((1, 1), (1, 1), (0, 0)), mode="edge")
    gx = (p[:-2, 2:] + 2 * p[1:-1, 2:] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[1:-1, :-2] + p[2:, :-2])
    gy = (p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) - (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:])
    mag_c = np.abs(gx) + np.abs(gy)
    best = np.argmax(mag_c, axis=2)                              # first channel with the largest magnitude
    ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    m, dx, dy = mag_c[ii, jj, best], gx[ii, jj, best], gy[ii, jj, best]
    mp = np.pad(m, 1)                                            # zero outside the image
    ctr = mp[1:-1, 1:-1]
    ax, ay = np.abs(dx).astype(np.int64), np.abs(dy).astype(np.int64) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = (~horiz) & (ay > tg67)
    diag = ~(horiz | vert)
    same = (dx ^ dy) >= 0                                        # gradient along the main diagonal
    k_h = (ctr > mp[1:-1, :-2]) & (ctr >= mp[1:-1, 2:])
    k_v = (ctr > mp[:-2, 1:-1]) & (ctr >= mp[2:, 1:-1])
    k_d1 = (ctr > mp[:-2, :-2]) & (ctr > mp[2:, 2:])
    k_d2 = (ctr > mp[:-2, 2:]) & (ctr > mp[2:, :-2])
    keep = (m > low) & ((horiz & k_h) | (vert & k_v) | (diag & same & k_d1) | (diag & ~same & k_d2))
    strong = keep & (m > high)
    lab, _ = ndimage.label(keep, structure=np.ones((3, 3), int))    # hysteresis = candidates 8-connected to a strong pixel
    good = np.unique(lab[strong])
    return np.where(np.isin(lab, good[good > 0]), 255, 0).astype(np.uint8)


# ------------------------------------------------------------------ the detector and the dataset's control video
def canny_detector(frame, low=100, high=200, resolution=512):
    """controlnet_aux.CannyDetector()(frame): uint8 [H,W,3] -> uint8 [H',W',3]."""
    if frame.ndim == 2:
        frame = frame[:, :, None]
    if frame.shape[2] == 1:
        frame = np.concatenate([frame] * 3, axis=2)              # HWC3
    h, w, _ = frame.shape
    hd, wd, k = resize_target(h, w, resolution)
    if (hd, wd) == (h, w):
        img = frame
    elif k > 1:
        img = resize_lanczos4_u8(frame, hd, wd)
    else:
        img = resize_area_u8(frame, hd, wd)
    e = canny_u8(img, low, high)
    return np.stack([e, e, e], axis=2)                           # HWC3; the INTER_LINEAR resize to the same size is a copy


def control_video(frames):
    """DS:559-578: uint8 [T,H,W,3] -> float32 [T,H,W,3] holding the bf16 values (x / 127.5 - 1 rounded to bf16)."""
    import torch
    out = []
    for f in np.asarray(frames):
        c = canny_detector(f)
        if c.shape[:2] != f.shape[:2]:
            c = resize_area_u8(c, f.shape[0], f.shape[1])
        out.append(c)
    t = torch.from_numpy(np.stack(out, 0)).to(torch.float32) / 127.5 - 1.0
    return t.to(torch.bfloat16)
