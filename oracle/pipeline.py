"""oracle/pipeline.py -- TEST INFRASTRUCTURE: CPU fp32 restatement of the inference driver.

Follows propainter_inference.py: get_ref_index :36-58, compute_flow :61-99, complete_flow
:102-156, image_propagation :159-225, feature_propagation :228-311, process_inpainting :314-341.
"""
from __future__ import annotations

import numpy as np
import torch

from . import generator as G
from . import raft as R
from . import rfc as C


def get_ref_index(mid, neighbor_ids, length, ref_stride, ref_num):
    if ref_num == -1:
        return [i for i in range(0, length, ref_stride) if i not in neighbor_ids]
    out = []
    lo = max(0, mid - ref_stride * (ref_num // 2))
    hi = min(length, mid + ref_stride * (ref_num // 2))
    for i in range(lo, hi, ref_stride):
        if i not in neighbor_ids:
            if len(out) > ref_num:
                break
            out.append(i)
    return out


def compute_flow(sd_raft, frames, iters):
    """Per-pair RAFT results do not depend on the reference's clip chunking (:65-90: instance norm
    is per sample, batch norm is in eval mode), so all pairs are evaluated pair by pair."""
    t = frames.shape[1]
    ff, fb = [], []
    for i in range(t - 1):
        a, b = R.raft_bidirectional(sd_raft, frames[:, i:i + 2], iters)
        ff.append(a)
        fb.append(b)
    return torch.cat(ff, 1), torch.cat(fb, 1)


def complete_flow(sd_rfc, flows, flow_masks, subvideo_length):
    n = flows[0].shape[1]
    if n <= subvideo_length:
        pred = C.forward_bidirect_flow(sd_rfc, flows, flow_masks)
        return C.combine_flow(flows, pred, flow_masks)
    pad = 5
    outs_f, outs_b = [], []
    for f in range(0, n, subvideo_length):
        s, e = max(0, f - pad), min(n, f + subvideo_length + pad)
        ps, pe = f - s, e - min(n, f + subvideo_length)
        sub = (flows[0][:, s:e], flows[1][:, s:e])
        m = flow_masks[:, s:e + 1]
        pred = C.combine_flow(sub, C.forward_bidirect_flow(sd_rfc, sub, m), m)
        outs_f.append(pred[0][:, ps:e - s - pe])
        outs_b.append(pred[1][:, ps:e - s - pe])
    return torch.cat(outs_f, 1), torch.cat(outs_b, 1)


def image_propagation(frames, masks_dilated, flows, subvideo_length):
    t = frames.shape[1]
    masked = frames * (1 - masks_dilated)
    sub = min(100, subvideo_length)
    if t <= sub:
        prop, upd = G.image_propagation(masked, flows[0], flows[1], masks_dilated, "nearest")
        return frames * (1 - masks_dilated) + prop * masks_dilated, upd
    pad = 10
    fr, mk = [], []
    for f in range(0, t, sub):
        s, e = max(0, f - pad), min(t, f + sub + pad)
        ps, pe = f - s, e - min(t, f + sub)
        prop, upd = G.image_propagation(masked[:, s:e], flows[0][:, s:e - 1], flows[1][:, s:e - 1],
                                        masks_dilated[:, s:e], "nearest")
        uf = frames[:, s:e] * (1 - masks_dilated[:, s:e]) + prop * masks_dilated[:, s:e]
        fr.append(uf[:, ps:e - s - pe])
        mk.append(upd[:, ps:e - s - pe])
    return torch.cat(fr, 1), torch.cat(mk, 1)


def window_schedule(length, neighbor_length, ref_stride, subvideo_length):
    ns = neighbor_length // 2
    ref_num = subvideo_length // ref_stride if length > subvideo_length else -1
    sched = []
    for f in range(0, length, ns):
        nb = list(range(max(0, f - ns), min(length, f + ns + 1)))
        sched.append((nb, get_ref_index(f, nb, length, ref_stride, ref_num)))
    return sched


def compose_window(composed, pred_img, masks_dilated, original_frames, neighbor_ids):
    """:283-307 -- uint8 compose + order-dependent 0.5/0.5 blending (truncating casts)."""
    pred = ((pred_img + 1) / 2).permute(0, 2, 3, 1).numpy() * 255
    bm = masks_dilated[0, neighbor_ids].permute(0, 2, 3, 1).numpy().astype(np.uint8)
    for i, idx in enumerate(neighbor_ids):
        img = np.array(pred[i]).astype(np.uint8) * bm[i] + original_frames[idx] * (1 - bm[i])
        if composed[idx] is None:
            composed[idx] = img
        else:
            composed[idx] = composed[idx].astype(np.float32) * 0.5 + img.astype(np.float32) * 0.5
        composed[idx] = composed[idx].astype(np.uint8)


def feature_propagation(sd_gen, updated_frames, updated_masks, masks_dilated, flows, original_frames, neighbor_length,
                        ref_stride, subvideo_length, return_pred=False):
    t = updated_frames.shape[1]
    composed = [None] * t
    preds = []
    for nb, refs in window_schedule(t, neighbor_length, ref_stride, subvideo_length):
        ids = nb + refs
        pred = G.generator_forward(sd_gen, updated_frames[:, ids], (flows[0][:, nb[:-1]], flows[1][:, nb[:-1]]),
                                   masks_dilated[:, ids], updated_masks[:, ids], len(nb))[0]
        preds.append(pred)
        compose_window(composed, pred, masks_dilated, original_frames, nb)
    return (composed, preds) if return_pred else composed


def run(sds, frames, flow_masks, masks_dilated, original_frames, *, raft_iter, neighbor_length, ref_stride,
        subvideo_length, return_trace=False):
    """process_inpainting + feature_propagation on prepared tensors (all fp32 CPU)."""
    with torch.no_grad():
        gt = compute_flow(sds["raft"], frames, raft_iter)
        pred = complete_flow(sds["rfc"], gt, flow_masks, subvideo_length)
        uf, um = image_propagation(frames, masks_dilated, pred, subvideo_length)
        comp, preds = feature_propagation(sds["gen"], uf, um, masks_dilated, pred, original_frames, neighbor_length,
                                          ref_stride, subvideo_length, return_pred=True)
    if return_trace:
        return comp, {"gt_flows": gt, "pred_flows": pred, "updated_frames": uf, "updated_masks": um, "pred_imgs": preds}
    return comp
