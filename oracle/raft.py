"""oracle/raft.py -- TEST INFRASTRUCTURE: CPU fp32 restatement of the reference's RAFT ("basic",
test_mode) used for bidirectional flow.  NCHW torch functional ops, weights read straight
from the reference state-dict layout (keys without the `module.` prefix are accepted too).

Follows: model/modules/RAFT/raft.py:94-152 (forward), :81-92 (convex upsample),
corr.py:12-60 (volume, pyramid, lookup), update.py:6-154 (update block),
extractor.py:5-57,121-193 (encoders), utils/utils.py:66-86; flow_comp_raft.py:39-58 (RAFT_bi).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _sd(sd, prefix):
    pre = "module." + prefix if any(k.startswith("module.") for k in sd) else prefix
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


def _norm(x, p, name, kind):
    if kind == "instance":  # extractor.py:32-36,134-135 -- no affine, per-sample statistics
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"],
                        p[name + ".bias"], training=False, eps=1e-5)


def _res_block(x, p, pre, kind, stride):
    y = F.relu(_norm(F.conv2d(x, p[pre + "conv1.weight"], p[pre + "conv1.bias"], stride=stride, padding=1), p,
                     pre + "norm1", kind))
    y = F.relu(_norm(F.conv2d(y, p[pre + "conv2.weight"], p[pre + "conv2.bias"], padding=1), p, pre + "norm2", kind))
    if stride != 1:
        x = F.conv2d(x, p[pre + "downsample.0.weight"], p[pre + "downsample.0.bias"], stride=stride)
        x = _norm(x, p, pre + "norm3", kind)
    return F.relu(x + y)


def encoder(x, p, kind):
    """BasicEncoder (extractor.py:170-193): /8 resolution, 256 channels."""
    x = F.relu(_norm(F.conv2d(x, p["conv1.weight"], p["conv1.bias"], stride=2, padding=3), p, "norm1", kind))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _res_block(x, p, f"{layer}.0.", kind, stride)
        x = _res_block(x, p, f"{layer}.1.", kind, 1)
    return F.conv2d(x, p["conv2.weight"], p["conv2.bias"])


def corr_pyramid(f1, f2, levels=4):
    """All-pairs volume / sqrt(C) and its avg-pooled pyramid (corr.py:12-27,52-60)."""
    n, c, h, w = f1.shape
    vol = torch.matmul(f1.view(n, c, h * w).transpose(1, 2), f2.view(n, c, h * w)) / (c ** 0.5)
    vol = vol.reshape(n * h * w, 1, h, w)
    pyr = [vol]
    for _ in range(levels - 1):
        vol = F.avg_pool2d(vol, 2, stride=2)
        pyr.append(vol)
    return pyr


def corr_lookup(pyr, coords, radius=4):
    """Bilinear 9x9 window lookup per level (corr.py:29-50); channel = l*81 + i*9 + j where the
    first window index i offsets X and j offsets Y."""
    n, _, h, w = coords.shape
    c = coords.permute(0, 2, 3, 1).reshape(n * h * w, 1, 1, 2)
    d = torch.arange(-radius, radius + 1, dtype=coords.dtype)
    # delta[i, j] = (d[i], d[j]) added to (x, y)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), -1).view(1, 2 * radius + 1, 2 * radius + 1, 2)
    out = []
    for lvl, vol in enumerate(pyr):
        pos = c / (2 ** lvl) + delta
        hh, ww = vol.shape[-2:]
        gx = 2 * pos[..., 0] / (ww - 1) - 1
        gy = 2 * pos[..., 1] / (hh - 1) - 1
        s = F.grid_sample(vol, torch.stack((gx, gy), -1), align_corners=True)
        out.append(s.view(n, h, w, -1))
    return torch.cat(out, -1).permute(0, 3, 1, 2).contiguous()


def update_block(p, net, inp, corr, flow):
    """BasicUpdateBlock (update.py:145-154) -> (net, 0.25*mask, delta_flow)."""
    def cv(name, x, pad):
        return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], padding=pad)

    cor = F.relu(cv("encoder.convc1", corr, 0))
    cor = F.relu(cv("encoder.convc2", cor, 1))
    flo = F.relu(cv("encoder.convf1", flow, 3))
    flo = F.relu(cv("encoder.convf2", flo, 1))
    mot = torch.cat([F.relu(cv("encoder.conv", torch.cat([cor, flo], 1), 1)), flow], 1)
    x = torch.cat([inp, mot], 1)
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([net, x], 1)
        z = torch.sigmoid(cv("gru.convz" + sfx, hx, pad))
        r = torch.sigmoid(cv("gru.convr" + sfx, hx, pad))
        q = torch.tanh(cv("gru.convq" + sfx, torch.cat([r * net, x], 1), pad))
        net = (1 - z) * net + z * q
    delta = cv("flow_head.conv2", F.relu(cv("flow_head.conv1", net, 1)), 1)
    mask = 0.25 * cv("mask.2", F.relu(cv("mask.0", net, 1)), 0)
    return net, mask, delta


def convex_upsample(flow, mask):
    """raft.py:81-92."""
    n, _, h, w = flow.shape
    m = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    nb = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    up = torch.sum(m * nb, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * w)


def coords_grid(n, h, w):
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(n, 1, 1, 1)


def raft_forward(sd, image1, image2, iters, return_trace=False):
    """RAFT.forward(test_mode=True): returns the 8x-upsampled flow of the last iteration."""
    fnet, cnet, upd = _sd(sd, "fnet."), _sd(sd, "cnet."), _sd(sd, "update_block.")
    n = image1.shape[0]
    fm = encoder(torch.cat([image1, image2], 0), fnet, "instance")
    f1, f2 = fm[:n], fm[n:]
    pyr = corr_pyramid(f1, f2)
    ctx = encoder(image1, cnet, "batch")
    net, inp = torch.tanh(ctx[:, :128]), torch.relu(ctx[:, 128:])
    h8, w8 = image1.shape[2] // 8, image1.shape[3] // 8
    c0 = coords_grid(n, h8, w8)
    c1 = c0.clone()
    trace = {"fmap1": f1, "fmap2": f2, "net0": net, "inp": inp, "corr0": None}
    mask = None
    for it in range(iters):
        corr = corr_lookup(pyr, c1)
        if it == 0:
            trace["corr0"] = corr
        net, mask, delta = update_block(upd, net, inp, corr, c1 - c0)
        c1 = c1 + delta
    flow_up = convex_upsample(c1 - c0, mask)
    if return_trace:
        trace.update({"flow_lr": c1 - c0, "mask": mask, "net": net})
        return flow_up, trace
    return flow_up


def raft_bidirectional(sd, frames, iters):
    """RAFT_bi.forward (flow_comp_raft.py:39-58). frames [1,T,3,H,W] in [-1,1] ->
    (flows_forward, flows_backward), each [1,T-1,2,H,W]."""
    b, t, c, h, w = frames.shape
    a = frames[:, :-1].reshape(-1, c, h, w)
    bb = frames[:, 1:].reshape(-1, c, h, w)
    ff = raft_forward(sd, a, bb, iters).view(b, t - 1, 2, h, w)
    fb = raft_forward(sd, bb, a, iters).view(b, t - 1, 2, h, w)
    return ff, fb
