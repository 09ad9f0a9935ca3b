"""Seeded synthetic inputs: a smooth random texture translated by a bounded
sinusoidal global motion, plus a static centre-third box mask.  Used by bench.py, smoke() and the
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
