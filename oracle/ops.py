"""oracle/ops.py -- TEST INFRASTRUCTURE: third-party operator restatements.

torchvision.ops.deform_conv2d is NOT in /root/reference (unpinned dependency);
it is restated here from its documented contract and anchored on the reference's two call sites
(recurrent_flow_completion.py:44-53, propainter.py:73-82).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def deform_conv2d(x, offset, weight, bias=None, stride=1, padding=0, dilation=1, mask=None):
    """Modulated deformable convolution (DCNv2), torchvision contract.

    x [N,Cin,H,W]; offset [N, 2*dg*K, Ho, Wo] with channel g*2K + 2k = dy, +1 = dx for tap
    k = i*kw + j; mask [N, dg*K, Ho, Wo]; sample position
    (ho*s - p + i*d + dy, wo*s - p + j*d + dx); bilinear, each out-of-range corner contributes 0;
    input channels split evenly over dg groups; out = W . (mask * sample) + b.
    """
    def pair(v):
        return (v, v) if isinstance(v, int) else tuple(v)

    sh, sw = pair(stride)
    ph, pw = pair(padding)
    dh, dw = pair(dilation)
    n, cin, h, w = x.shape
    cout, cin_w, kh, kw = weight.shape
    assert cin_w == cin, "weight groups != 1 not needed on this path"
    K = kh * kw
    ho, wo = offset.shape[-2:]
    dg = offset.shape[1] // (2 * K)
    cg = cin // dg
    ys = torch.arange(ho, dtype=x.dtype, device=x.device).view(1, ho, 1) * sh - ph
    xs = torch.arange(wo, dtype=x.dtype, device=x.device).view(1, 1, wo) * sw - pw
    cols = x.new_zeros(n, cin, K, ho, wo)
    for g in range(dg):
        xg = x[:, g * cg:(g + 1) * cg]
        for k in range(K):
            i, j = divmod(k, kw)
            dy = offset[:, g * 2 * K + 2 * k]
            dx = offset[:, g * 2 * K + 2 * k + 1]
            py = ys + i * dh + dy
            px = xs + j * dw + dx
            # align_corners=True grid_sample with zeros padding == per-corner zero contribution
            gx = 2.0 * px / max(w - 1, 1) - 1.0
            gy = 2.0 * py / max(h - 1, 1) - 1.0
            samp = F.grid_sample(xg, torch.stack((gx, gy), -1), mode="bilinear", padding_mode="zeros",
                                 align_corners=True)
            if mask is not None:
                samp = samp * mask[:, g * K + k].unsqueeze(1)
            cols[:, g * cg:(g + 1) * cg, k] = samp
    out = torch.einsum("nckhw,ock->nohw", cols, weight.reshape(cout, cin, K))
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
