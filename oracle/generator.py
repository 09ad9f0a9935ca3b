"""oracle/generator.py -- TEST INFRASTRUCTURE: CPU fp32 restatement of InpaintGenerator inference.

Follows model/propainter.py: fbConsistencyCheck :27-36, DeformableAlignment :62-82,
BidirectionalPropagation :118-231, Encoder :261-275, InpaintGenerator.forward :358-453,
img_propagation :350-356; model/modules/flow_loss_utils.py:6-51 (flow_warp);
model/modules/sparse_transformer.py: SoftSplit :18-36, SoftComp :50-64, FusionFeedForward
:79-123, SparseWindowAttention :201-393 (restated per SURVEY.md 9.13 as an explicit
key-set construction rather than roll/cat/index), blocks :413-467.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .ops import deform_conv2d


# --------------------------------------------------------------------------- warping
def flow_warp(x, flow, mode="bilinear"):
    """x [n,c,h,w], flow [n,h,w,2] (dx,dy) in pixels; zeros padding, align_corners=True."""
    n, c, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack((gx, gy), 2).to(x.dtype) + flow
    nx = 2.0 * grid[..., 0] / max(w - 1, 1) - 1.0
    ny = 2.0 * grid[..., 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((nx, ny), 3), mode=mode, padding_mode="zeros", align_corners=True)


def fb_check(flow_fw, flow_bw, a1=0.01, a2=0.5):
    bw = flow_warp(flow_bw, flow_fw.permute(0, 2, 3, 1))
    diff = flow_fw + bw
    mag = (flow_fw ** 2).sum(1, keepdim=True) + (bw ** 2).sum(1, keepdim=True)
    return ((diff ** 2).sum(1, keepdim=True) < a1 * mag + a2).to(flow_fw.dtype)


def _bin(m, th=0.1):
    return (m > th).to(m.dtype)


# --------------------------------------------------------------------------- propagation
def image_propagation(frames, flows_f, flows_b, masks, mode="nearest"):
    """BidirectionalPropagation(learnable=False) (:118-231). frames [b,t,3,h,w], masks [b,t,1,h,w],
    flows [b,t-1,2,h,w]. Returns (prop_frames [b,t,3,h,w], prop_masks [b,t,1,h,w]) of the forward pass."""
    b, t, c, h, w = frames.shape
    feats = {"input": [frames[:, i] for i in range(t)]}
    msks = {"input": [masks[:, i] for i in range(t)]}
    src = "input"
    for name in ("backward_1", "forward_1"):
        if name == "backward_1":
            order, fidx, f_prop, f_chk = list(range(t))[::-1], list(range(t))[::-1], flows_f, flows_b
        else:
            order, fidx, f_prop, f_chk = list(range(t)), list(range(-1, t - 1)), flows_b, flows_f
        fo, mo = [], []
        for i, idx in enumerate(order):
            cur, mcur = feats[src][idx], msks[src][idx]
            if i == 0:
                fp, mp = cur, mcur
            else:
                fl = f_prop[:, fidx[i]]
                valid = fb_check(fl, f_chk[:, fidx[i]])
                warped = flow_warp(fp, fl.permute(0, 2, 3, 1), mode)
                mvalid = _bin(flow_warp(mp, fl.permute(0, 2, 3, 1)))
                union = _bin(mcur * valid * (1 - mvalid))
                fp = union * warped + (1 - union) * cur
                mp = _bin(mcur * (1 - valid * (1 - mvalid)))
            fo.append(fp)
            mo.append(mp)
        if name == "backward_1":
            fo, mo = fo[::-1], mo[::-1]
        feats[name], msks[name] = fo, mo
        src = name
    return torch.stack(feats["forward_1"], 1), torch.stack(msks["forward_1"], 1)


def _deform_align(p, pre, x, cond, flow, max_mag=3.0):
    o = cond
    for i in (0, 2, 4):
        o = F.leaky_relu(F.conv2d(o, p[f"{pre}conv_offset.{i}.weight"], p[f"{pre}conv_offset.{i}.bias"], padding=1), 0.1)
    o = F.conv2d(o, p[pre + "conv_offset.6.weight"], p[pre + "conv_offset.6.bias"], padding=1)
    o1, o2, msk = torch.chunk(o, 3, dim=1)
    offset = max_mag * torch.tanh(torch.cat((o1, o2), 1))
    offset = offset + flow.flip(1).repeat(1, offset.size(1) // 2, 1, 1)
    return deform_conv2d(x, offset, p[pre + "weight"], p[pre + "bias"], 1, 1, 1, torch.sigmoid(msk))


def feature_propagation(p, x, flows_f, flows_b, mask):
    """BidirectionalPropagation(128, learnable=True). x [b,t,128,h,w], mask [b,t,2,h,w]."""
    b, t, c, h, w = x.shape
    pre = "feat_prop_module."
    feats = {"input": [x[:, i] for i in range(t)]}
    mlist = [mask[:, i] for i in range(t)]
    src = "input"
    for name in ("backward_1", "forward_1"):
        if name == "backward_1":
            order, fidx, f_prop, f_chk = list(range(t))[::-1], list(range(t))[::-1], flows_f, flows_b
        else:
            order, fidx, f_prop, f_chk = list(range(t)), list(range(-1, t - 1)), flows_b, flows_f
        fo = []
        for i, idx in enumerate(order):
            cur, mcur = feats[src][idx], mlist[idx]
            if i == 0:
                fp = cur
            else:
                fl = f_prop[:, fidx[i]]
                valid = fb_check(fl, f_chk[:, fidx[i]])
                warped = flow_warp(fp, fl.permute(0, 2, 3, 1), "bilinear")
                cond = torch.cat([cur, warped, fl, valid, mcur], 1)
                fp = _deform_align(p, f"{pre}deform_align.{name}.", fp, cond, fl)
            bb = F.conv2d(torch.cat([cur, fp, mcur], 1), p[f"{pre}backbone.{name}.0.weight"],
                          p[f"{pre}backbone.{name}.0.bias"], padding=1)
            bb = F.conv2d(F.leaky_relu(bb, 0.2), p[f"{pre}backbone.{name}.2.weight"], p[f"{pre}backbone.{name}.2.bias"],
                          padding=1)
            fp = fp + bb
            fo.append(fp)
        feats[name] = fo[::-1] if name == "backward_1" else fo
        src = name
    ob = torch.stack(feats["backward_1"], 1).view(-1, c, h, w)
    of = torch.stack(feats["forward_1"], 1).view(-1, c, h, w)
    fu = F.conv2d(torch.cat([ob, of, mask.view(-1, 2, h, w)], 1), p[pre + "fuse.0.weight"], p[pre + "fuse.0.bias"], padding=1)
    fu = F.conv2d(F.leaky_relu(fu, 0.2), p[pre + "fuse.2.weight"], p[pre + "fuse.2.bias"], padding=1)
    return (fu + x.view(-1, c, h, w)).view(b, t, c, h, w)


# --------------------------------------------------------------------------- encoder / decoder
ENC_GROUPS = {10: 2, 12: 4, 14: 8, 16: 1}


def encoder(p, x):
    """Encoder.forward (:261-275) incl. the group-interleaved skip concat."""
    out = x
    x0 = None
    for i in range(0, 18, 2):
        if i == 8:
            x0 = out
        if i > 8:
            g = ENC_GROUPS[i]
            bt, _, h, w = out.shape
            out = torch.cat([x0.view(bt, g, -1, h, w), out.view(bt, g, -1, h, w)], 2).view(bt, -1, h, w)
        stride = 2 if i in (0, 4) else 1
        out = F.leaky_relu(F.conv2d(out, p[f"encoder.layers.{i}.weight"], p[f"encoder.layers.{i}.bias"], stride=stride,
                                    padding=1, groups=ENC_GROUPS.get(i, 1)), 0.2)
    return out


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def decoder(p, x):
    x = F.leaky_relu(F.conv2d(_up2(x), p["decoder.0.conv.weight"], p["decoder.0.conv.bias"], padding=1), 0.2)
    x = F.leaky_relu(F.conv2d(x, p["decoder.2.weight"], p["decoder.2.bias"], padding=1), 0.2)
    x = F.leaky_relu(F.conv2d(_up2(x), p["decoder.4.conv.weight"], p["decoder.4.conv.bias"], padding=1), 0.2)
    return F.conv2d(x, p["decoder.6.weight"], p["decoder.6.bias"], padding=1)


# --------------------------------------------------------------------------- transformer
def soft_split(p, x):
    """SoftSplit == conv k7 s3 p3. x [bt,128,h,w] -> tokens [bt, fh, fw, 512]."""
    w = p["ss.embedding.weight"].view(512, 128, 7, 7)
    return F.conv2d(x, w, p["ss.embedding.bias"], stride=3, padding=3).permute(0, 2, 3, 1)


def soft_comp(p, tok, hw):
    """SoftComp: Linear -> fold (overlap-add) -> 3x3 bias_conv. tok [bt, fh, fw, 512]."""
    bt = tok.shape[0]
    f = F.linear(tok.reshape(bt, -1, 512), p["sc.embedding.weight"], p["sc.embedding.bias"])
    f = F.fold(f.permute(0, 2, 1), output_size=hw, kernel_size=(7, 7), stride=(3, 3), padding=(3, 3))
    return F.conv2d(f, p["sc.bias_conv.weight"], p["sc.bias_conv.bias"], padding=1)


def fusion_ffn(p, pre, x, hw):
    """FusionFeedForward (:79-123). x [bt, n_vecs, 512]."""
    bt, n, _ = x.shape
    y = F.linear(x, p[pre + "fc1.0.weight"], p[pre + "fc1.0.bias"])
    kw = dict(output_size=hw, kernel_size=(7, 7), stride=(3, 3), padding=(3, 3))
    norm = F.fold(torch.ones(1, 49, n), **kw)
    y = F.fold(y.permute(0, 2, 1), **kw) / norm
    y = F.unfold(y, kernel_size=(7, 7), stride=(3, 3), padding=(3, 3)).permute(0, 2, 1)
    return F.linear(F.gelu(y), p[pre + "fc2.1.weight"], p[pre + "fc2.1.bias"])


def window_attention(p, pre, x, mask, t_ind, window=(5, 9), heads=4):
    """SparseWindowAttention.forward. x [1,t,h,w,512] (already LayerNorm-ed), mask [1,l_t,h,w,1]."""
    b, t, h, w, c = x.shape
    assert b == 1
    wh, ww = window
    nh, nw = math.ceil(h / wh), math.ceil(w / ww)
    H, W = nh * wh, nw * ww
    x = F.pad(x, (0, 0, 0, W - w, 0, H - h))
    mask = F.pad(mask, (0, 0, 0, W - w, 0, H - h))
    q = F.linear(x, p[pre + "query.weight"], p[pre + "query.bias"])[0]
    k = F.linear(x, p[pre + "key.weight"], p[pre + "key.bias"])[0]
    v = F.linear(x, p[pre + "value.weight"], p[pre + "value.bias"])[0]
    pool = F.conv2d(x[0].permute(0, 3, 1, 2), p[pre + "pool_layer.weight"], p[pre + "pool_layer.bias"], stride=4,
                    groups=c).permute(0, 2, 3, 1)
    pk = F.linear(pool, p[pre + "key.weight"], p[pre + "key.bias"]).reshape(t, -1, c)
    pv = F.linear(pool, p[pre + "value.weight"], p[pre + "value.bias"]).reshape(t, -1, c)
    eh, ew = (wh + 1) // 2, (ww + 1) // 2
    wmask = F.max_pool2d(mask[0, :, :, :, 0], (wh, ww), (wh, ww)).sum(0)  # [nh, nw]
    out = torch.zeros(t, H, W, c)
    d = c // heads
    scale = 1.0 / math.sqrt(d)
    # neighbour offsets relative to the window origin (rolled windows minus the window itself)
    nb = [(dr, dc) for dr in list(range(-eh, wh - eh)) + list(range(eh, wh + eh))
          for dc in list(range(-ew, ww - ew)) + list(range(ew, ww + ew))]
    nb = sorted({o for o in nb if not (0 <= o[0] < wh and 0 <= o[1] < ww)})
    # only the 4 diagonal rolls exist: rows from one vertical roll AND cols from one horizontal roll
    for wi in range(nh):
        for wj in range(nw):
            r0, c0 = wi * wh, wj * ww
            rows = torch.arange(r0, r0 + wh)
            cols = torch.arange(c0, c0 + ww)
            qw = q[:, rows][:, :, cols].reshape(t, wh * ww, heads, d)
            kw_ = k[:, rows][:, :, cols].reshape(t, wh * ww, heads, d)
            vw = v[:, rows][:, :, cols].reshape(t, wh * ww, heads, d)
            if wmask[wi, wj] > 0:
                rr = torch.tensor([(r0 + o[0]) % H for o in nb])
                cc = torch.tensor([(c0 + o[1]) % W for o in nb])
                kn = k[:, rr, cc].reshape(t, len(nb), heads, d)
                vn = v[:, rr, cc].reshape(t, len(nb), heads, d)
                kk = torch.cat([kw_, kn, pk.view(t, -1, heads, d)], 1)[t_ind].reshape(-1, heads, d)
                vv = torch.cat([vw, vn, pv.view(t, -1, heads, d)], 1)[t_ind].reshape(-1, heads, d)
                qq = qw.reshape(t * wh * ww, heads, d)
                att = torch.softmax(torch.einsum("qhd,khd->hqk", qq, kk) * scale, -1)
                y = torch.einsum("hqk,khd->qhd", att, vv).reshape(t, wh, ww, c)
            else:
                att = torch.softmax(torch.einsum("tqhd,tkhd->thqk", qw, kw_) * scale, -1)
                y = torch.einsum("thqk,tkhd->tqhd", att, vw).reshape(t, wh, ww, c)
            out[:, r0:r0 + wh, c0:c0 + ww] = y
    out = out[:, :h, :w]
    return F.linear(out, p[pre + "proj.weight"], p[pre + "proj.bias"])[None]


def transformer(p, tok, hw, lmask, depths=8, t_dilation=2):
    """TemporalSparseTransformerBlock.forward. tok [1,t,fh,fw,512], lmask [1,l_t,fh,fw,1]."""
    t = tok.shape[1]
    x = tok
    for i in range(depths):
        pre = f"transformers.transformer.{i}."
        t_ind = torch.arange(i % t_dilation, t, t_dilation)
        y = F.layer_norm(x, (512,), p[pre + "norm1.weight"], p[pre + "norm1.bias"])
        x = x + window_attention(p, pre + "attention.", y, lmask, t_ind)
        y = F.layer_norm(x, (512,), p[pre + "norm2.weight"], p[pre + "norm2.bias"])
        b_, t_, fh, fw, c = y.shape
        x = x + fusion_ffn(p, pre + "mlp.", y.view(t_, fh * fw, c), hw).view(b_, t_, fh, fw, c)
    return x


# --------------------------------------------------------------------------- full generator
def generator_forward(p, frames, flows_bi, masks_in, masks_updated, l_t, return_trace=False):
    """InpaintGenerator.forward (eval). frames [1,t,3,H,W]; flows [1,l_t-1,2,H,W]; masks [1,t,1,H,W]."""
    b, t, _, H, W = frames.shape
    enc = encoder(p, torch.cat([frames.view(t, 3, H, W), masks_in.view(t, 1, H, W), masks_updated.view(t, 1, H, W)], 1))
    _, c, h, w = enc.shape
    enc = enc.view(1, t, c, h, w)
    local, ref = enc[:, :l_t], enc[:, l_t:]
    ds_f = F.interpolate(flows_bi[0].view(-1, 2, H, W), scale_factor=0.25, mode="bilinear",
                         align_corners=False).view(1, l_t - 1, 2, h, w) / 4.0
    ds_b = F.interpolate(flows_bi[1].view(-1, 2, H, W), scale_factor=0.25, mode="bilinear",
                         align_corners=False).view(1, l_t - 1, 2, h, w) / 4.0
    ds_min = F.interpolate(masks_in.reshape(-1, 1, H, W), scale_factor=0.25, mode="nearest").view(1, t, 1, h, w)
    ds_mup = F.interpolate(masks_updated[:, :l_t].reshape(-1, 1, H, W), scale_factor=0.25,
                           mode="nearest").view(1, l_t, 1, h, w)
    mpool = F.max_pool2d(ds_min[:, :l_t].reshape(-1, 1, h, w), 7, 3, 3)
    mpool = mpool.view(1, l_t, 1, *mpool.shape[-2:]).permute(0, 1, 3, 4, 2)
    prop_mask = torch.cat([ds_min[:, :l_t], ds_mup], 2)
    local_p = feature_propagation(p, local, ds_f, ds_b, prop_mask)
    enc2 = torch.cat((local_p, ref), 1)
    tok = soft_split(p, enc2.view(-1, c, h, w))[None]
    tok_out = transformer(p, tok, (h, w), mpool)
    trans = soft_comp(p, tok_out[0], (h, w)).view(1, t, c, h, w)
    enc3 = enc2 + trans
    out = torch.tanh(decoder(p, enc3[0, :l_t])).view(1, l_t, 3, H, W)
    if return_trace:
        return out, {"enc": enc, "local_prop": local_p, "tok": tok, "tok_out": tok_out, "enc3": enc3}
    return out
