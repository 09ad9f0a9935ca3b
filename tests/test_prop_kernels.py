"""Flow-guided propagation kernels against the oracle (fp32 CPU restatement of the reference)."""
import pytest
import torch
import torch.nn.functional as F

from comfyui_propainter_nodes_amd import imgprop, ops
from oracle import generator as OG


def _smooth_flows(g, t, h, w, mag):
    f = torch.randn(2, t, 2, h // 8 + 1, w // 8 + 1, generator=g) * mag
    return F.interpolate(f.view(-1, 2, h // 8 + 1, w // 8 + 1), size=(h, w), mode="bilinear", align_corners=True).view(2, t, 2, h, w)


def test_image_propagation_bit_exact(backend):
    """Image propagation is select/copy work: frames, and the propagated masks, must match exactly."""
    dev = backend
    g = torch.Generator().manual_seed(21)
    T, H, W = 5, 40, 56
    frames = torch.rand(T, H, W, 3, generator=g) * 2 - 1
    masks = torch.zeros(T, H, W, dtype=torch.uint8)
    masks[:, 10:30, 15:40] = 1
    masks[2, 5:12, 40:50] = 1
    flows = _smooth_flows(g, T - 1, H, W, 4.0)  # [2,T-1,2,H,W]
    fl_nhwc = flows.permute(0, 1, 3, 4, 2).contiguous()
    pf, pm = imgprop.image_propagation(frames.to(dev), masks.to(dev), fl_nhwc.to(dev))
    m = masks.float()[None, :, None]
    fr = frames.permute(0, 3, 1, 2)[None]
    rf, rm = OG.image_propagation(fr * (1 - m), flows[0][None], flows[1][None], m, "nearest")
    assert torch.equal(pm.cpu().float(), rm[0, :, 0])
    assert torch.equal(pf.cpu(), rf[0].permute(0, 2, 3, 1))


@pytest.mark.parametrize("dt", [torch.float16, torch.float32], ids=["f16", "f32"])
def test_flow_down4_aux_and_warp(backend, dt):
    dev = backend
    g = torch.Generator().manual_seed(22)
    n, H, W = 3, 32, 48
    flows = _smooth_flows(g, n, H, W, 6.0)
    ff = flows[0].permute(0, 2, 3, 1).contiguous()
    fb = flows[1].permute(0, 2, 3, 1).contiguous()
    d = torch.empty(n, H // 4, W // 4, 2, device=dev)
    ops.flow_down4(ff.to(dev), d)
    ref = F.interpolate(flows[0], scale_factor=0.25, mode="bilinear", align_corners=False) / 4.0
    assert torch.allclose(d.cpu(), ref.permute(0, 2, 3, 1), atol=1e-6)
    # fb-check planes + bilinear feature warp at 1/4 resolution
    h, w = H // 4, W // 4
    dsf, dsb = ref, F.interpolate(flows[1], scale_factor=0.25, mode="bilinear", align_corners=False) / 4.0
    mp = torch.zeros(n, h, w, 8, dtype=dt)
    mp[..., 0] = (torch.rand(n, h, w, generator=g) > 0.5).to(dt)
    mp[..., 1] = (torch.rand(n, h, w, generator=g) > 0.5).to(dt)
    aux = torch.empty(n, h, w, 8, device=dev, dtype=dt)
    a = dsf.permute(0, 2, 3, 1).contiguous()
    b = dsb.permute(0, 2, 3, 1).contiguous()
    ops.featprop_aux(a.to(dev), b.to(dev), mp.to(dev), aux)
    valid = OG.fb_check(dsf, dsb)
    got = aux.float().cpu()
    assert torch.equal(got[..., 2], valid[:, 0])
    assert torch.allclose(got[..., 0:2], a.to(dt).float()) and torch.equal(got[..., 3:5], mp[..., 0:2].float())
    x = torch.randn(n, h, w, 16, generator=g)
    out = torch.empty(n, h, w, 16, device=dev)
    ops.flow_warp(x.to(dev), a.to(dev), out)
    refw = OG.flow_warp(x.permute(0, 3, 1, 2), a, "bilinear").permute(0, 2, 3, 1)
    assert torch.allclose(out.cpu(), refw, atol=1e-5)


@pytest.mark.parametrize("dt", [torch.float16, torch.float32], ids=["f16", "f32"])
def test_pack_encoder_input(backend, dt):
    dev = backend
    g = torch.Generator().manual_seed(23)
    T, H, W = 2, 8, 8
    frames = torch.rand(T, H, W, 3, generator=g) * 2 - 1
    prop = torch.rand(T, H, W, 3, generator=g) * 2 - 1
    m_in = (torch.rand(T, H, W, generator=g) > 0.5).to(torch.uint8)
    m_up = (torch.rand(T, H, W, generator=g) > 0.5).to(torch.uint8)
    out = torch.empty(T, H, W, 8, device=dev, dtype=dt)
    upd = torch.empty(T, H, W, 3, device=dev)
    ops.pack_encoder_input(frames.to(dev), prop.to(dev), m_in.to(dev), m_up.to(dev), out, upd)
    m = m_in.float()[..., None]
    ref = frames * (1 - m) + prop * m
    assert torch.equal(upd.cpu(), ref)
    got = out.float().cpu()
    assert torch.equal(got[..., :3], ref.to(dt).float()) and torch.equal(got[..., 3], m_in.float()) and torch.equal(got[..., 4], m_up.float())
    assert torch.all(got[..., 5:] == 0)
