"""RAFT-stage gather/normalisation kernels against the oracle's torch formulation (fp32, atol 2e-5)."""
import pytest
import torch
import torch.nn.functional as F

from comfyui_propainter_nodes_amd import ops
from oracle import raft as OR


def test_im2col_matches_unfold(backend):
    dev = backend
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 11, 13, 3, generator=g)
    out = torch.empty(2, 6, 7, 160, device=dev)
    ops.im2col(x.to(dev), out, 7, 7, stride=2, padding=3)
    ref = F.unfold(x.permute(0, 3, 1, 2), 7, padding=3, stride=2)  # [n, c*49, L] with (c,ky,kx) order
    ref = ref.view(2, 3, 49, 6, 7).permute(0, 3, 4, 2, 1).reshape(2, 6, 7, 147)
    assert torch.equal(out.cpu()[..., :147], ref)
    assert torch.all(out.cpu()[..., 147:] == 0)
    # replicate padding + channel-slice input (flow stored inside a wider buffer)
    buf = torch.randn(1, 9, 10, 8, generator=g)
    out2 = torch.empty(1, 5, 5, 96, device=dev)
    ops.im2col(buf.to(dev)[..., 5:8], out2, 5, 5, stride=2, padding=2, pad_mode="replicate")
    xp = F.pad(buf[..., 5:8].permute(0, 3, 1, 2), (2, 2, 2, 2), mode="replicate")
    ref2 = F.unfold(xp, 5, stride=2).view(1, 3, 25, 5, 5).permute(0, 3, 4, 2, 1).reshape(1, 5, 5, 75)
    assert torch.equal(out2.cpu()[..., :75], ref2)


@pytest.mark.parametrize("C", [64, 96, 128, 256, 30])  # (30: not a multiple of 4 -> the scalar form of the kernels)
def test_instnorm(backend, C):
    dev = backend
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 24, 31, C, generator=g) * 3 + 1.5
    skip = torch.randn(2, 24, 31, C, generator=g)
    ref = F.instance_norm(x.permute(0, 3, 1, 2), eps=1e-5).permute(0, 2, 3, 1)
    y = torch.empty_like(x, device=dev)
    ops.instnorm(x.to(dev), y, relu_pre=True)
    assert torch.allclose(y.cpu(), F.relu(ref), atol=2e-5)
    ops.instnorm(x.to(dev), y, relu_pre=True, skip=skip.to(dev), relu_post=True)
    assert torch.allclose(y.cpu(), F.relu(skip + F.relu(ref)), atol=2e-5)
    xin = x.to(dev)
    ops.instnorm(xin, xin)  # in place, no relu
    assert torch.allclose(xin.cpu(), ref, atol=2e-5)


def test_pyramid_lookup_upsample(backend):
    dev = backend
    g = torch.Generator().manual_seed(3)
    n, h, w = 2, 16, 18
    f1 = torch.randn(n, 32, h, w, generator=g)
    f2 = torch.randn(n, 32, h, w, generator=g)
    pyr_ref = OR.corr_pyramid(f1, f2)
    vol = pyr_ref[0].view(n, h * w, h, w).contiguous().to(dev)
    pyr = [vol]
    for lvl in range(3):
        ph, pw = pyr[-1].shape[2] // 2, pyr[-1].shape[3] // 2
        nxt = torch.empty(n, h * w, ph, pw, device=dev)
        ops.avgpool2x2(pyr[-1].view(n * h * w, *pyr[-1].shape[2:]), nxt.view(n * h * w, ph, pw))
        assert torch.allclose(nxt.cpu().view(-1, 1, ph, pw), pyr_ref[lvl + 1], atol=1e-6)
        pyr.append(nxt)
    # flows large enough to leave the image on some pixels (zero padding path)
    flow = torch.randn(n, h, w, 2, generator=g) * 6
    buf = torch.zeros(n, h, w, 8, device=dev)
    buf[..., 6:8] = flow.to(dev)
    out = torch.empty(n, h, w, 324, device=dev)
    ops.corr_lookup(pyr, buf[..., 6:8], out)
    coords = OR.coords_grid(n, h, w) + flow.permute(0, 3, 1, 2)
    ref = OR.corr_lookup(pyr_ref, coords).permute(0, 2, 3, 1)
    assert torch.allclose(out.cpu(), ref, atol=2e-5)
    # the same pyramid with levels 0 and 1 in 4 x 8 tiles (r03 layout; 18 columns -> 3 tile columns with padding, 16 rows ->
    # 4 tile rows; level 1: 8 x 9 -> 2 x 2 tiles): pooled tiled -> tiled -> row-major, and an identical lookup, bit for bit
    order = torch.tensor(ops.tiled_order(h, w))
    volz = torch.cat([vol.cpu().view(n, h * w, h * w), torch.zeros(n, h * w, 1)], 2)
    t0 = volz.index_select(2, order).contiguous().to(dev)                    # [n, hw, tiled pitch]
    tp = [(t0, h, w, True)]
    for lvl in range(1, 4):
        _, hi, wi, ti = tp[-1]
        ho, wo, to = hi // 2, wi // 2, lvl == 1
        nxt = torch.empty(n, h * w, ops.tiled_pitch(ho, wo) if to else ho * wo, device=dev)
        ops.avgpool2x2(tp[-1][0].view(n * h * w, -1), nxt.view(n * h * w, -1), hw=(hi, wi), in_tiled=ti, out_tiled=to)
        tp.append((nxt, ho, wo, to))
    o1 = torch.tensor(ops.tiled_order(h // 2, w // 2))
    lvl1 = torch.cat([pyr[1].cpu().view(n, h * w, -1), torch.zeros(n, h * w, 1)], 2).index_select(2, o1)
    assert torch.equal(tp[1][0].cpu(), lvl1) and torch.equal(tp[2][0].cpu().view(-1), pyr[2].cpu().view(-1))
    out_t = torch.empty(n, h, w, 324, device=dev)
    ops.corr_lookup(tp, buf[..., 6:8], out_t)
    assert torch.equal(out_t.cpu(), out.cpu())
    # convex upsampling
    mask = torch.randn(n, h, w, 576, generator=g)
    up = torch.empty(n, 8 * h, 8 * w, 2, device=dev)
    ops.convex_upsample(mask.to(dev), buf[..., 6:8], up)
    ref_up = OR.convex_upsample(flow.permute(0, 3, 1, 2), mask.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert torch.allclose(up.cpu(), ref_up, atol=2e-5 * ref_up.abs().max().item())


def test_lookup_fused_with_projection(backend):
    """r06, pp_corr_lookup_conv: the correlation lookup fused with RAFT's 324 -> 256 projection (convc1 + relu) against the two
    launches it replaces -- pp_corr_lookup, then the 1x1 PP_F32X2 convolution over its output -- and against float64 torch on the
    lookup's own values: tiled and row-major pyramids, flows that leave the image, a pixel count that is not a multiple of the
    32-pixel tile (2 x 17 x 19 = 646 pixels: 21 tiles, the last one ragged), the flow as a channel view of a wider buffer."""
    dev = backend
    g = torch.Generator().manual_seed(11)
    n, h, w = 2, 17, 19
    hw = h * w
    vol = torch.randn(n, hw, h, w, generator=g).to(dev)
    pyr = [vol]
    for lvl in range(3):
        ph, pw = pyr[-1].shape[2] // 2, pyr[-1].shape[3] // 2
        nxt = torch.empty(n, hw, ph, pw, device=dev)
        ops.avgpool2x2(pyr[-1].view(n * hw, *pyr[-1].shape[2:]), nxt.view(n * hw, ph, pw))
        pyr.append(nxt)
    order = torch.tensor(ops.tiled_order(h, w))
    t0 = torch.cat([vol.cpu().view(n, hw, hw), torch.zeros(n, hw, 1)], 2).index_select(2, order).contiguous().to(dev)
    tp = [(t0, h, w, True)]
    for lvl in range(1, 4):
        _, hi, wi, ti = tp[-1]
        ho, wo, to = hi // 2, wi // 2, lvl == 1
        nxt = torch.empty(n, hw, ops.tiled_pitch(ho, wo) if to else ho * wo, device=dev)
        ops.avgpool2x2(tp[-1][0].view(n * hw, -1), nxt.view(n * hw, -1), hw=(hi, wi), in_tiled=ti, out_tiled=to)
        tp.append((nxt, ho, wo, to))
    flow = torch.randn(n, h, w, 2, generator=g) * 5
    buf = torch.zeros(n, h, w, 128, device=dev)
    buf[..., 126:128] = flow.to(dev)
    fl = buf[..., 126:128]
    wt = torch.randn(256, 324, 1, 1, generator=g) * 0.05
    b = torch.randn(256, generator=g)
    spec = ops.make_conv_spec(wt, b, torch.float32, split=True).to(dev)
    for pyramid in (pyr, tp):
        corr = torch.empty(n, h, w, 352, device=dev)[..., :324]
        ops.corr_lookup(pyramid, fl, corr)
        two = torch.empty(n, h, w, 256, device=dev)
        ops.conv2d(spec, [corr], two, act="relu")
        one = torch.full((n, h, w, 256), 7.0, device=dev)
        ops.corr_lookup_conv(pyramid, fl, spec, one, act="relu")
        ref = F.relu(corr.double().cpu() @ wt.view(256, 324).double().t() + b.double())
        scale = max(1.0, ref.abs().max().item())
        assert (one.double().cpu() - ref).abs().max().item() < 2e-5 * scale
        assert (one - two).abs().max().item() < 1e-5 * scale      # same products, another summation order
    # a fused epilogue on the one-launch form (out = relu(...) + aux), and the refusal of any other projection
    aux = torch.randn(n, h, w, 256, generator=g).to(dev)
    one2 = torch.empty(n, h, w, 256, device=dev)
    ops.corr_lookup_conv(tp, fl, spec, one2, act="relu", epi="add", aux1=aux)
    assert (one2 - (one + aux)).abs().max().item() < 1e-5 * scale
    bad = ops.make_conv_spec(torch.randn(128, 324, 1, 1), None, torch.float32, split=True).to(dev)
    with pytest.raises(ValueError):
        ops.corr_lookup_conv(tp, fl, bad, torch.empty(n, h, w, 128, device=dev))
