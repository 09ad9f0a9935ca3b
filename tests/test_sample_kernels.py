"""Sampling kernels against their torch formulation (oracle/ops.py restates torchvision's deform_conv2d)."""
import pytest
import torch
import torch.nn.functional as F

from comfyui_propainter_nodes_amd import ops
from oracle.ops import deform_conv2d


def test_deform_conv_matches_contract(backend):
    """pp_deform_cols + GEMM == deform_conv2d(x=[x0|x1], offset, W, b, mask) for both call-site shapes."""
    dev = backend
    g = torch.Generator().manual_seed(11)
    for c0, c1, with_flow in ((32, 32, False), (32, 0, True)):
        n, h, w, dg = 2, 9, 11, 16
        cin = c0 + c1
        x = torch.randn(n, h, w, cin, generator=g)
        off = torch.randn(n, h, w, 2 * dg * 9, generator=g) * 2.5  # some samples leave the image
        msk = torch.rand(n, h, w, dg * 9, generator=g)
        flow = torch.randn(n, h, w, 2, generator=g) * 2 if with_flow else None
        wgt = torch.randn(24, cin, 3, 3, generator=g) * 0.1
        bias = torch.randn(24, generator=g)
        om = torch.cat([off, msk], -1).to(dev)
        cols = torch.empty(n, h, w, 9 * cin, device=dev, dtype=torch.float16)
        xh = x.half().to(dev)
        ops.deform_cols(xh[..., :c0], xh[..., c0:] if c1 else None, om, cols, dg=dg,
                        flow=flow.to(dev) if with_flow else None)
        spec = ops.make_conv_spec(wgt.permute(0, 2, 3, 1).reshape(24, 9 * cin, 1, 1), bias, torch.float16).to(dev)
        out = torch.empty(n, h, w, 24, device=dev, dtype=torch.float16)
        ops.conv2d(spec, [cols], out)
        offr = off.clone()
        if with_flow:  # propainter.py:67-68: offset + flow.flip(1) tiled -> (dy += flow_y, dx += flow_x)
            offr = offr + torch.stack([flow[..., 1], flow[..., 0]], -1).repeat(1, 1, 1, dg * 9)
        ref = deform_conv2d(x.half().float().permute(0, 3, 1, 2), offr.permute(0, 3, 1, 2), wgt.half().float(), bias, 1, 1, 1,
                            msk.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        err = (out.float().cpu() - ref).abs().max().item()
        assert err < 4e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("xcd", ["1", "0"])
def test_fused_deform_conv_matches_the_two_kernel_form(backend, pp_knobs, xcd):
    """pp_deform_conv (sampling feeds the MFMA operand, no column tensor) == pp_deform_cols + 1x1 pp_conv2d: the sampled f16
    values are the same bit for bit, so the two differ by fp32 summation order at most (not at all where the 1x1 convolution
    is the split-K kernel, whose order pp_deform_conv follows); also against the torchvision contract (oracle/ops.py).
    Both call-site shapes (two inputs / one input + flow), partial pixel tiles, offsets that leave the image and non-finite
    offsets, pixel blocks in XCD-contiguous and in launch order, f16 and f32 output with a fused epilogue."""
    dev = backend
    pp_knobs(PP_DEFORM_XCD=xcd)
    g = torch.Generator().manual_seed(21)
    for c0, c1, with_flow, odt, (h, w) in ((64, 64, False, torch.float16, (9, 11)), (32, 0, True, torch.float32, (7, 5))):
        n, dg, cout = 2, 4 if c1 == 0 else 16, 128 if c1 else 40
        cin = c0 + c1
        x = torch.randn(n, h, w, cin, generator=g)
        off = torch.randn(n, h, w, 2 * dg * 9, generator=g) * 2.5
        off[0, 1, 2, 5] = float("nan")
        off[1, 3, 1, 7] = float("inf")
        msk = torch.rand(n, h, w, dg * 9, generator=g)
        flow = torch.randn(n, h, w, 2, generator=g) * 2 if with_flow else None
        wgt = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
        bias = torch.randn(cout, generator=g)
        om = torch.cat([off, msk], -1).to(dev)
        xh = x.half().to(dev)
        x0, x1 = xh[..., :c0], (xh[..., c0:] if c1 else None)
        fl = flow.to(dev) if with_flow else None
        spec = ops.make_conv_spec(wgt.permute(0, 2, 3, 1).reshape(cout, 9 * cin, 1, 1), bias, torch.float16).to(dev)
        res = torch.randn(n, h, w, cout, generator=g).to(odt).to(dev)
        cols = torch.empty(n, h, w, 9 * cin, device=dev, dtype=torch.float16)
        ops.deform_cols(x0, x1, om, cols, dg=dg, flow=fl)
        two = ops.conv2d(spec, [cols], torch.empty(n, h, w, cout, device=dev, dtype=odt), act="leaky", act_param=0.1, epi="add", aux1=res)
        one = ops.deform_conv(spec, x0, x1, om, torch.empty(n, h, w, cout, device=dev, dtype=odt), dg=dg, flow=fl, act="leaky",
                              act_param=0.1, epi="add", aux1=res)
        scale = max(1.0, two.float().abs().max().item())
        assert (one.float() - two.float()).abs().max().item() < (2e-3 if odt == torch.float16 else 2e-5) * scale
        if c1:   # 9 x 128 channels: the 1x1 convolution is the split-K kernel, whose summation order pp_deform_conv follows
            assert torch.equal(one, two)
        offr = torch.nan_to_num(off, nan=-1e8, posinf=-1e8, neginf=-1e8)   # non-finite offsets sample nothing
        if with_flow:
            offr = offr + torch.stack([flow[..., 1], flow[..., 0]], -1).repeat(1, 1, 1, dg * 9)
        ref = deform_conv2d(x.half().float().permute(0, 3, 1, 2), offr.permute(0, 3, 1, 2), wgt.half().float(), bias, 1, 1, 1,
                            msk.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        ref = F.leaky_relu(ref, 0.1) + res.float().cpu()
        err = (one.float().cpu() - ref).abs().max().item()
        assert err < 4e-3 * max(1.0, ref.abs().max().item()), err


def test_upsample2x_align_corners(backend, pp_knobs):
    """Against torch's bilinear interpolation; r06: the 2 x 2-block kernel (one thread per input pixel: 9 loads per 4 stores) must
    equal the per-output-pixel kernel BIT FOR BIT -- odd sizes, a single row / column (H - 1 = 0: every output row reads row 0), the
    decoder's aspect, and 4 channels (the per-element kernel: no 8-channel pieces)."""
    dev = backend
    g = torch.Generator().manual_seed(12)
    for shape in ((2, 7, 9, 16), (1, 1, 5, 8), (1, 6, 1, 24), (1, 45, 80, 8), (1, 3, 4, 4)):
        x = torch.randn(*shape, generator=g)
        n, h, w, c = shape
        for dt, tol in ((torch.float32, 1e-5), (torch.float16, 2e-3)):
            outs = []
            for b4 in ("1", "0"):
                pp_knobs(PP_UPSAMPLE_B4=b4)
                out = torch.full((n, 2 * h, 2 * w, c), float("nan"), device=dev, dtype=dt)
                ops.upsample2x(x.to(dt).to(dev), out)
                outs.append(out.cpu())
            assert torch.equal(outs[0], outs[1]), shape
            ref = F.interpolate(x.to(dt).float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear",
                                align_corners=True).permute(0, 2, 3, 1)
            assert (outs[0].float() - ref).abs().max().item() < tol * max(1e-6, ref.abs().max().item())


def test_rfc_prep_and_combine(backend):
    dev = backend
    g = torch.Generator().manual_seed(13)
    T, H, W = 3, 6, 8
    flows = torch.randn(2, T, H, W, 2, generator=g)
    masks = (torch.rand(T + 1, H, W, generator=g) > 0.5).to(torch.uint8)
    x = torch.empty(T, 2, H, W, 4, device=dev, dtype=torch.float16)
    ops.rfc_prep(flows.to(dev), masks.to(dev), x)
    m = masks.float()
    ref_f = torch.cat([flows[0] * (1 - m[:-1, ..., None]), m[:-1, ..., None]], -1)
    ref_b = torch.flip(torch.cat([flows[1] * (1 - m[1:, ..., None]), m[1:, ..., None]], -1), dims=[0])
    got = x.float().cpu()
    assert torch.allclose(got[:, 0, ..., :3], ref_f.half().float()) and torch.allclose(got[:, 1, ..., :3], ref_b.half().float())
    pred = torch.randn(T, 2, H, W, 2, generator=g).half()
    out = torch.empty(2, T, H, W, 2, device=dev)
    ops.flow_combine(pred.to(dev), flows.to(dev), masks.to(dev), out)
    pf = pred[:, 0].float()
    pb = torch.flip(pred[:, 1].float(), dims=[0])
    ref0 = pf * m[:-1, ..., None] + flows[0] * (1 - m[:-1, ..., None])
    ref1 = pb * m[1:, ..., None] + flows[1] * (1 - m[1:, ..., None])
    assert torch.allclose(out.cpu()[0], ref0) and torch.allclose(out.cpu()[1], ref1)


def test_deform_conv_hand_computed_samples(backend):
    """An independent pin of the deform_conv2d contract (oracle, fixtures and kernel all descend from
    oracle/ops.py).  The expected numbers below are worked out BY HAND from torchvision's documented rule -- sample position
    (y - pad + i + dy, x - pad + j + dx), bilinear, a sample with h <= -1, h >= H, w <= -1 or w >= W is 0 and every
    out-of-range corner contributes 0 -- on an image that is linear in (y, x), so an interior sample is the linear function
    itself.  Checked for the oracle AND for pp_deform_cols + pp_conv2d."""
    dev = backend
    n, h, w, cin, dg, K = 1, 6, 7, 32, 16, 9
    ci, tap = 3, 4                      # channel 3 -> offset group 3 // (32 / 16) = 1; tap 4 = (i, j) = (1, 1): base position (y, x)
    grp = ci // (cin // dg)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    x = torch.zeros(n, h, w, cin)
    x[0, :, :, ci] = 4.0 * (10.0 * yy + xx)          # f(y, x) = 4 (10 y + x): exact in f16 on this grid
    off = torch.zeros(n, h, w, 2 * dg * K)
    msk = torch.ones(n, h, w, dg * K)
    dy_c, dx_c, m_c = grp * 2 * K + 2 * tap, grp * 2 * K + 2 * tap + 1, grp * K + tap
    cases = [  # (y, x, dy, dx, mask, expected)
        (2, 3, 0.5, 0.25, 1.0, 4.0 * (10 * 2.5 + 3.25)),           # interior: the linear function at (2.5, 3.25) = 113
        (0, 2, -0.5, 0.0, 1.0, 0.5 * 4.0 * 2.0),                   # h = -0.5: row -1 contributes 0, row 0 with weight 0.5 -> 4
        (0, 1, -1.0, 0.0, 1.0, 0.0),                               # h = -1 exactly: outside
        (1, w - 1, 0.0, 0.5, 1.0, 0.5 * 4.0 * (10 + (w - 1))),     # w = W - 0.5: column W contributes 0 -> 32
        (3, w - 1, 0.0, 1.0, 1.0, 0.0),                            # w = W exactly: outside
        (4, 4, 0.0, 0.0, 0.25, 0.25 * 4.0 * 44.0),                 # modulation mask -> 44
        (5, 0, 0.75, -0.5, 1.0, 0.25 * 0.5 * 4.0 * 50.0),          # h = 5.75, w = -0.5: only corner (5, 0), weight 0.25 * 0.5 -> 25
    ]
    for (y, xq, dy, dx, m, _) in cases:
        off[0, y, xq, dy_c], off[0, y, xq, dx_c], msk[0, y, xq, m_c] = dy, dx, m
    wgt = torch.zeros(8, cin, 3, 3)
    wgt[0, ci, 1, 1] = 1.0              # output channel 0 = mask * sample of channel `ci` at tap (1, 1)
    ref = deform_conv2d(x.permute(0, 3, 1, 2), off.permute(0, 3, 1, 2), wgt, None, 1, 1, 1, msk.permute(0, 3, 1, 2))[0, 0]
    cols = torch.empty(n, h, w, 9 * cin, device=dev, dtype=torch.float16)
    ops.deform_cols(x.half().to(dev), None, torch.cat([off, msk], -1).to(dev), cols, dg=dg)
    spec = ops.make_conv_spec(wgt.permute(0, 2, 3, 1).reshape(8, 9 * cin, 1, 1), None, torch.float16).to(dev)
    out = torch.empty(n, h, w, 8, device=dev, dtype=torch.float16)
    ops.conv2d(spec, [cols], out)
    got = out[0, :, :, 0].float().cpu()
    for (y, xq, _, _, _, want) in cases:
        assert abs(ref[y, xq].item() - want) < 1e-4, ("oracle", y, xq, ref[y, xq].item(), want)
        assert abs(got[y, xq].item() - want) < 1e-2 + 1e-3 * abs(want), ("kernel", y, xq, got[y, xq].item(), want)
    # everywhere else the offsets are zero: the centre tap reproduces the image
    untouched = torch.ones(h, w, dtype=torch.bool)
    for (y, xq, *_rest) in cases:
        untouched[y, xq] = False
    assert torch.allclose(ref[untouched], x[0, :, :, ci][untouched]) and torch.allclose(got[untouched], x[0, :, :, ci][untouched])


@pytest.mark.gpu
def test_upsample2x_beyond_2_to_32_elements(hip_lib):
    """r04 regression (found by the cfg5_90f_node fixture, fp16 "disable"): HIP launches a grid as a 32-bit global size and silently
    TRUNCATES a launch of 2^32 threads or more.  The per-element fp32 upsample of flow completion's decoder -- 160 images of
    720x1280x32 = 4.7e9 outputs -- wrote only its first 14.6 images (completed flows 30 px off from frame 7 on).  The fp32 tensors
    now take the 8-channels-per-thread kernel; a launch that would still exceed the limit FAILS instead of returning garbage."""
    dev = torch.device("cuda:0")
    N, h, w, C = 160, 360, 640, 32                      # the tensor of the bug: 4.7e9 output elements (18.9 GB)
    x = torch.randn(2, h, w, C, device=dev).repeat(N // 2, 1, 1, 1).contiguous()   # images repeat with period 2
    out = torch.empty(N, 2 * h, 2 * w, C, device=dev)
    ops.upsample2x(x, out)
    torch.cuda.synchronize()
    assert torch.equal(out[N - 2], out[0]) and torch.equal(out[N - 1], out[1])      # the LAST images were written, and right
    ref = torch.nn.functional.interpolate(x[:1].permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    assert (out[0] - ref[0].permute(1, 2, 0)).abs().max().item() < 1e-5
    del out, x
    # the per-element kernel (C % 8 != 0) on 2^32 outputs: refused
    big = torch.empty(1, 32768, 32768, 4, device=dev)                                # 2^32 outputs (17 GB)
    src = torch.zeros(1, 16384, 16384, 4, device=dev)
    with pytest.raises(RuntimeError) as e:
        ops.upsample2x(src, big)
    assert "2^32" in str(e.value)
