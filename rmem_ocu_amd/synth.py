"""Synthetic clips for tests and bench.py.

Frames are low-frequency noise (a coarse Gaussian grid, bilinearly upsampled 8x)
seen through a window that drifts 2-3 px per frame, plus 0.05 * white noise; they
stand in for ImageNet-normalised RGB (evaluator input, dataloaders/video_transforms.py:
676-680).  The first-frame mask is K axis-aligned rectangles labelled 1..K.
Everything is a pure function of (seed, sizes) through numpy PCG64 streams.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def network_size(h: int, w: int, max_long: int = 1040, stride: int = 16, align_corners: bool = True, scale: float = 1.0):
    """Stride-aligned network input size, dataloaders/video_transforms.py:575-622
    (max_size branch; TEST_MAX_SIZE = 1.3 * 800 = 1040, tools/eval.py:128); ``scale`` is one entry of TEST_MULTISCALE."""
    sc = None
    if max(h, w) > max_long:
        sc = float(max_long) / max(h, w)
    nh, nw = (h, w) if sc is None else (sc * h, sc * w)
    nh, nw = int(nh * scale), int(nw * scale)
    if align_corners:
        if (nh - 1) % stride:
            nh = int(np.around((nh - 1) / stride) * stride + 1)
        if (nw - 1) % stride:
            nw = int(np.around((nw - 1) / stride) * stride + 1)
    else:
        if nh % stride:
            nh = int(np.around(nh / stride) * stride)
        if nw % stride:
            nw = int(np.around(nw / stride) * stride)
    return nh, nw


def make_clip(seed: int, num_frames: int, h: int, w: int, num_objs: int = 3):
    """Returns (frames float32 [n,3,h,w], first_mask int32 [1,1,h,w])."""
    rng = np.random.Generator(np.random.PCG64([seed, 0x5EED]))
    drift = rng.integers(2, 4, size=(num_frames, 2))            # 2-3 px per frame
    sign = rng.choice([-1, 1], size=2)
    off = np.cumsum(drift * sign, axis=0)
    off -= off.min(axis=0)
    ch, cw = h + int(off[:, 0].max()), w + int(off[:, 1].max())
    coarse = rng.standard_normal((1, 3, ch // 8 + 2, cw // 8 + 2)).astype(np.float32)
    canvas = F.interpolate(torch.from_numpy(coarse), size=(ch, cw), mode='bilinear', align_corners=True)[0]
    frames = torch.empty(num_frames, 3, h, w, dtype=torch.float32)
    for i in range(num_frames):
        y, x = int(off[i, 0]), int(off[i, 1])
        noise = torch.from_numpy(rng.standard_normal((3, h, w)).astype(np.float32))
        frames[i] = 1.5 * canvas[:, y:y + h, x:x + w] + 0.05 * noise
    mask = np.zeros((h, w), dtype=np.int32)
    for k in range(1, num_objs + 1):
        rh, rw = int(h * rng.uniform(0.15, 0.35)), int(w * rng.uniform(0.12, 0.3))
        y0, x0 = int(rng.integers(0, h - rh)), int(rng.integers(0, w - rw))
        mask[y0:y0 + rh, x0:x0 + rw] = k
    return frames, torch.from_numpy(mask)[None, None]


def clip_masks(seed: int, num_frames: int, h: int, w: int, num_objs: int = 3) -> torch.Tensor:
    """Ground-truth label maps [n, h, w] (int64) of make_clip(seed, ...): the rectangles are anchored to the drifting
    texture, so frame i shows them shifted by -(offset_i - offset_0).  Replays make_clip's random stream exactly."""
    rng = np.random.Generator(np.random.PCG64([seed, 0x5EED]))
    drift = rng.integers(2, 4, size=(num_frames, 2))
    sign = rng.choice([-1, 1], size=2)
    off = np.cumsum(drift * sign, axis=0)
    off -= off.min(axis=0)
    ch, cw = h + int(off[:, 0].max()), w + int(off[:, 1].max())
    rng.standard_normal((1, 3, ch // 8 + 2, cw // 8 + 2))
    for _ in range(num_frames):
        rng.standard_normal((3, h, w))
    rects = []
    for k in range(1, num_objs + 1):
        rh, rw = int(h * rng.uniform(0.15, 0.35)), int(w * rng.uniform(0.12, 0.3))
        y0, x0 = int(rng.integers(0, h - rh)), int(rng.integers(0, w - rw))
        rects.append((k, y0, x0, rh, rw))
    out = torch.zeros(num_frames, h, w, dtype=torch.int64)
    for i in range(num_frames):
        dy, dx = int(off[i, 0] - off[0, 0]), int(off[i, 1] - off[0, 1])
        for k, y0, x0, rh, rw in rects:
            ya, xa = max(y0 - dy, 0), max(x0 - dx, 0)
            yb, xb = min(y0 - dy + rh, h), min(x0 - dx + rw, w)
            if yb > ya and xb > xa:
                out[i, ya:yb, xa:xb] = k
    return out
