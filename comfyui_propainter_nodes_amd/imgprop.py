"""Image propagation (BidirectionalPropagation(learnable=False), propainter.py:118-231 via
InpaintGenerator.img_propagation :350-356) on the MI355X.

Pure gather/select work at full resolution: one fused kernel per time step (fb-consistency
check + nearest frame warp + bilinear mask warp + mask algebra), fp32 coordinates, u8 masks.
"""
from __future__ import annotations

import torch

from . import ops


def image_propagation(frames: torch.Tensor, masks_u8: torch.Tensor, flows: torch.Tensor):
    """frames fp32 [T,H,W,3] in [-1,1], masks u8 [T,H,W] (dilated masks), flows fp32 [2,T-1,H,W,2]
    (completed forward / backward flows) -> (prop_frames fp32 [T,H,W,3], prop_masks u8 [T,H,W]) of the
    forward pass (propainter.py:224).  The caller blends frames*(1-m) + prop*m."""
    T, H, W, _ = frames.shape
    dev = frames.device
    fb = torch.empty(T, H, W, 3, device=dev)
    mb = torch.empty(T, H, W, dtype=torch.uint8, device=dev)
    ff = torch.empty(T, H, W, 3, device=dev)
    mf = torch.empty(T, H, W, dtype=torch.uint8, device=dev)
    # backward pass over the masked input frames: flows_forward propagate, flows_backward check
    for i, idx in enumerate(range(T - 1, -1, -1)):
        if i == 0:
            ops.img_prop_step(frames[idx], masks_u8[idx], fb[idx], mb[idx], mask_input=True)
        else:
            ops.img_prop_step(frames[idx], masks_u8[idx], fb[idx], mb[idx], f_prev=fb[idx + 1], m_prev=mb[idx + 1],
                              flow_prop=flows[0, idx], flow_check=flows[1, idx], mask_input=True)
    # forward pass over the backward results: flows_backward propagate, flows_forward check
    for idx in range(T):
        if idx == 0:
            ops.img_prop_step(fb[idx], mb[idx], ff[idx], mf[idx])
        else:
            ops.img_prop_step(fb[idx], mb[idx], ff[idx], mf[idx], f_prev=ff[idx - 1], m_prev=mf[idx - 1],
                              flow_prop=flows[1, idx - 1], flow_check=flows[0, idx - 1])
    return ff, mf
