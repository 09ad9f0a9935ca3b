"""Seeded synthetic inputs: a smooth random texture translated by a bounded
sinusoidal global motion, plus a static centre-third box mask (or a per-frame moving mask).  Used by bench.py, smoke() and the
tests; there are no datasets in the build/bench environment."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def synthetic_clip(T: int, H: int, W: int, seed: int = 1234) -> tuple[torch.Tensor, torch.Tensor]:
    """Return (image [T,H,W,3] float32 in [0,1], mask [1,H,W] float32 {0,1}) in ComfyUI conventions."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(1, 3, H // 8 + 8, W // 8 + 8, generator=g)
    base = F.interpolate(base, size=(H + 32, W + 32), mode="bicubic", align_corners=False).clamp(0, 1)[0]
    frames = []
    for t in range(T):
        oy = 16 + round(12 * math.sin(2 * math.pi * t / 40))
        ox = 16 + round(14 * math.sin(2 * math.pi * t / 56 + 1))
        frames.append(base[:, oy:oy + H, ox:ox + W].permute(1, 2, 0))
    image = torch.stack(frames, 0).contiguous()
    mask = torch.zeros(1, H, W)
    mask[:, H // 3:2 * H // 3, W // 3:2 * W // 3] = 1.0
    return image, mask


def moving_mask(T: int, H: int, W: int) -> torch.Tensor:
    """Per-frame MASK [T,H,W] float32 {0,1}: a (H/4 x W/4) box on a smooth closed orbit plus a small second box that
    appears only on every third frame (the reference dilates each mask frame on its own, image_utils.py:142-175)."""
    mask = torch.zeros(T, H, W)
    bh, bw = H // 4, W // 4
    for t in range(T):
        cy = H // 2 + round(H / 6 * math.sin(2 * math.pi * t / 30))
        cx = W // 2 + round(W / 5 * math.cos(2 * math.pi * t / 45))
        y0, x0 = max(0, cy - bh // 2), max(0, cx - bw // 2)
        mask[t, y0:y0 + bh, x0:x0 + bw] = 1.0
        if t % 3 == 0:
            mask[t, H // 8:H // 8 + H // 10, W // 10 + 4 * t:W // 10 + 4 * t + W // 12] = 1.0
    return mask
