"""oracle/rfc.py -- TEST INFRASTRUCTURE: CPU fp32 restatement of RecurrentFlowCompleteNet inference.

Follows model/recurrent_flow_completion.py: forward :315-354 (encoder :238-277, decoder
:282-300), BidirectionalPropagation :77-143, SecondOrderDeformableAlignment :32-53,
P3DBlock :162-205, deconv :146-159, forward_bidirect_flow :356-387, combine_flow :389-400.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .ops import deform_conv2d


def _lrelu(x, s=0.2):
    return F.leaky_relu(x, s)


def _p3d(x, p, pre, stride):
    """P3DBlock: (1,3,3) conv + LeakyReLU(0.2), then (3,1,1) conv, dilation 2, pad 2 (no residual)."""
    y = _lrelu(F.conv3d(x, p[pre + "conv1.0.weight"], p[pre + "conv1.0.bias"], stride=(1, stride, stride),
                        padding=(0, 1, 1)))
    return F.conv3d(y, p[pre + "conv2.0.weight"], p[pre + "conv2.0.bias"], padding=(2, 0, 0), dilation=(2, 1, 1))


def encode(p, inputs):
    """inputs [b,3,t,h,w] -> (x /2, e1 /4, mid /8)."""
    x = F.pad(inputs, (2, 2, 2, 2, 0, 0), mode="replicate")
    x = _lrelu(F.conv3d(x, p["downsample.0.weight"], p["downsample.0.bias"], stride=(1, 2, 2)))
    e1 = _lrelu(_p3d(x, p, "encoder1.0.", 1))
    e1 = _lrelu(_p3d(e1, p, "encoder1.2.", 2))
    e2 = _lrelu(_p3d(e1, p, "encoder2.0.", 1))
    e2 = _lrelu(_p3d(e2, p, "encoder2.2.", 2))
    m = e2
    for idx, d in (("0", 3), ("2", 2), ("4", 1)):
        m = _lrelu(F.conv3d(m, p[f"mid_dilation.{idx}.weight"], p[f"mid_dilation.{idx}.bias"], padding=(0, d, d),
                            dilation=(1, d, d)))
    return x, e1, m


def _deform_align(p, pre, x, cond, max_mag=5.0):
    """SecondOrderDeformableAlignment.forward (:32-53)."""
    o = cond
    for i in (0, 2, 4):
        o = F.leaky_relu(F.conv2d(o, p[f"{pre}conv_offset.{i}.weight"], p[f"{pre}conv_offset.{i}.bias"], padding=1), 0.1)
    o = F.conv2d(o, p[pre + "conv_offset.6.weight"], p[pre + "conv_offset.6.bias"], padding=1)
    o1, o2, msk = torch.chunk(o, 3, dim=1)
    offset = max_mag * torch.tanh(torch.cat((o1, o2), 1))
    return deform_conv2d(x, offset, p[pre + "weight"], p[pre + "bias"], 1, 1, 1, torch.sigmoid(msk))


def propagate(p, feats):
    """BidirectionalPropagation.forward (:77-143). feats [b,t,c,h,w] -> [b,t,c,h,w]."""
    b, t, c, h, w = feats.shape
    pre = "feat_prop_module."
    spatial = [feats[:, i] for i in range(t)]
    done = {}
    for name in ("backward_", "forward_"):
        order = list(range(t))[::-1] if name == "backward_" else list(range(t))
        hist = []
        prop = feats.new_zeros(b, c, h, w)
        for i, idx in enumerate(order):
            cur = spatial[idx]
            if i > 0:
                n2 = hist[-2] if i > 1 else torch.zeros_like(prop)
                cond = torch.cat([prop, cur, n2], 1)
                prop = _deform_align(p, f"{pre}deform_align.{name}.", torch.cat([prop, n2], 1), cond)
            parts = [cur] + ([done["backward_"][idx]] if name == "forward_" else []) + [prop]
            bb = F.conv2d(torch.cat(parts, 1), p[f"{pre}backbone.{name}.0.weight"], p[f"{pre}backbone.{name}.0.bias"],
                          padding=1)
            bb = F.conv2d(F.leaky_relu(bb, 0.1), p[f"{pre}backbone.{name}.2.weight"], p[f"{pre}backbone.{name}.2.bias"],
                          padding=1)
            prop = prop + bb
            hist.append(prop)
        done[name] = hist[::-1] if name == "backward_" else hist
    outs = [F.conv2d(torch.cat([done["backward_"][i], done["forward_"][i]], 1), p[pre + "fusion.weight"],
                     p[pre + "fusion.bias"]) for i in range(t)]
    return torch.stack(outs, 1) + feats


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def decode(p, prop, e1):
    """decoder2 + skip, decoder1, upsample (:282-300,338-345). prop [bt,128,h/8,w/8], e1 [bt,64,h/4,w/4]."""
    d2 = _lrelu(F.conv2d(prop, p["decoder2.0.weight"], p["decoder2.0.bias"], padding=1))
    d2 = _lrelu(F.conv2d(_up2(d2), p["decoder2.2.conv.weight"], p["decoder2.2.conv.bias"], padding=1)) + e1
    d1 = _lrelu(F.conv2d(d2, p["decoder1.0.weight"], p["decoder1.0.bias"], padding=1))
    d1 = _lrelu(F.conv2d(_up2(d1), p["decoder1.2.conv.weight"], p["decoder1.2.conv.bias"], padding=1))
    u = _lrelu(F.conv2d(d1, p["upsample.0.weight"], p["upsample.0.bias"], padding=1))
    return F.conv2d(_up2(u), p["upsample.2.conv.weight"], p["upsample.2.conv.bias"], padding=1)


def rfc_forward(p, masked_flows, masks, return_trace=False):
    """RecurrentFlowCompleteNet.forward (eval). masked_flows [b,t,2,h,w], masks [b,t,1,h,w]."""
    b, t, _, h, w = masked_flows.shape
    inputs = torch.cat((masked_flows.permute(0, 2, 1, 3, 4), masks.permute(0, 2, 1, 3, 4)), 1)
    _, e1, mid = encode(p, inputs)
    mid = mid.permute(0, 2, 1, 3, 4)
    prop = propagate(p, mid).reshape(-1, 128, h // 8, w // 8)
    e1f = e1.permute(0, 2, 1, 3, 4).reshape(-1, e1.shape[1], e1.shape[3], e1.shape[4])
    flow = decode(p, prop, e1f).view(b, t, 2, h, w)
    if return_trace:
        return flow, {"mid": mid, "prop": prop, "e1": e1f}
    return flow


def forward_bidirect_flow(p, flows_bi, masks):
    """:356-387 -- masks [b,t,1,h,w] for t frames, flows [b,t-1,2,h,w]."""
    mf, mb = masks[:, :-1], masks[:, 1:]
    pf = rfc_forward(p, flows_bi[0] * (1 - mf), mf)
    pb = rfc_forward(p, torch.flip(flows_bi[1] * (1 - mb), dims=[1]), torch.flip(mb, dims=[1]))
    return pf, torch.flip(pb, dims=[1])


def combine_flow(flows_bi, pred_bi, masks):
    """:389-400."""
    mf, mb = masks[:, :-1], masks[:, 1:]
    return pred_bi[0] * mf + flows_bi[0] * (1 - mf), pred_bi[1] * mb + flows_bi[1] * (1 - mb)
