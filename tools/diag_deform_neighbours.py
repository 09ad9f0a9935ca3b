"""r06 diagnostic (continues tools/diag_featprop_race.py): deform_cols gives WRONG values (host recomputation agrees with the solo
run) when `conv 3x3 128 -> 432, f32 output` runs next to it on a second stream.  Which property of the neighbour matters?"""
import os
import sys

import torch

sys.path.insert(0, '.')
os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = "1"
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
nw, h, w = 7, 90, 160
prop = torch.randn(nw, h, w, 128, generator=g).half().to(dev)
om = torch.cat([torch.randn(nw, h, w, 288, generator=g) * 3, torch.rand(nw, h, w, 144, generator=g)], 3).to(dev)
flow = (torch.randn(nw, h, w, 2, generator=g) * 2).to(dev)
cols = torch.empty(nw, h, w, 9 * 128, device=dev, dtype=torch.float16)
ops.deform_cols(prop, None, om, cols, flow=flow)
torch.cuda.synchronize()
ref = cols.clone()
side = torch.cuda.Stream(dev)
x16 = torch.randn(nw, h, w, 128, generator=g).half().to(dev)
x32 = torch.randn(nw, h, w, 128, generator=g).to(dev)


def conv(cout, k, odt, xin=x16, split=False, n=nw):
    wt = torch.randn(cout, 128, k, k, generator=g) * 0.03
    sp = ops.make_conv_spec(wt, torch.randn(cout, generator=g), xin.dtype, padding=k // 2, split=split).to(dev)
    out = torch.empty(n, h, w, cout, device=dev, dtype=odt)
    return lambda: ops.conv2d(sp, [xin[:n]], out, act="tanh")


big = torch.empty(nw, h, w, 432, device=dev)
cases = {
    "3x3 128->432 f16 in, f32 out (the offender)": conv(432, 3, torch.float32),
    "3x3 128->432 f16 in, f16 out": conv(432, 3, torch.float16),
    "3x3 128->128 f16 in, f32 out": conv(128, 3, torch.float32),
    "3x3 128->384 f16 in, f32 out": conv(384, 3, torch.float32),
    "1x1 128->432 f16 in, f32 out": conv(432, 1, torch.float32),
    "3x3 128->256 PP_F32X2 (f32 in / out)": conv(256, 3, torch.float32, x32, True),
    "torch: 174 MB f32 tensor * 1.0001 (elementwise)": lambda: big.mul_(1.0001),
    "torch: 174 MB f32 fill": lambda: big.fill_(1.0),
    "deform_cols itself on other buffers": None,
}
cols2 = torch.empty_like(cols)
prop2 = prop.clone()
om2 = om.clone()
flow2 = flow.clone()
cases["deform_cols itself on other buffers"] = lambda: ops.deform_cols(prop2, None, om2, cols2, flow=flow2)
for name, fn in cases.items():
    fn()
    torch.cuda.synchronize()
    nbad, cnt = 0, []
    for it in range(12):
        cols.fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4):
                fn()
        ops.deform_cols(prop, None, om, cols, flow=flow)
        torch.cuda.synchronize()
        d = cols != ref
        if bool(d.any()):
            nbad += 1
            cnt.append(int(d.sum()))
    print(f"{name:55s}: deform_cols wrong in {nbad:2d} of 12 runs {cnt[:4]}", flush=True)

# ---- other victims next to the offender: fp32 deform_cols, the 2x upsampling kernel, flow_warp
offender = cases["3x3 128->432 f16 in, f32 out (the offender)"]
prop32, cols32 = prop.float(), torch.empty(nw, h, w, 9 * 128, device=dev)
up_in, up_out = torch.randn(nw, h, w, 128, generator=g).half().to(dev), torch.empty(nw, 2 * h, 2 * w, 128, device=dev, dtype=torch.float16)
warp_out = torch.empty_like(prop)
victims = {
    "deform_cols<float>": (lambda: ops.deform_cols(prop32, None, om, cols32, flow=flow), cols32),
    "upsample2x (f16, 8 channels per thread)": (lambda: ops.upsample2x(up_in, up_out), up_out),
    "flow_warp (f16)": (lambda: ops.flow_warp(prop, flow, warp_out), warp_out),
    "deform_cols<half> with dg = 1 (one group of 128 channels)": None,
}
om1 = torch.cat([torch.randn(nw, h, w, 18, generator=g) * 3, torch.rand(nw, h, w, 9, generator=g)], 3).to(dev)
cols1 = torch.empty_like(cols)
victims["deform_cols<half> with dg = 1 (one group of 128 channels)"] = (lambda: ops.deform_cols(prop, None, om1, cols1, dg=1, flow=flow), cols1)
for name, (fn, out) in victims.items():
    fn()
    torch.cuda.synchronize()
    r = out.clone()
    nbad, cnt = 0, []
    for it in range(12):
        out.fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4):
                offender()
        fn()
        torch.cuda.synchronize()
        d = out != r
        if bool(d.any()):
            nbad += 1
            cnt.append(int(d.sum()))
    print(f"victim {name:58s}: wrong in {nbad:2d} of 12 runs {cnt[:4]}", flush=True)

# ---- is it the access pattern?  torch's own gather with deform_cols's index pattern into `om` (4-byte loads, 72 bytes apart)
npix = nw * h * w
pix = torch.arange(npix, device=dev).view(-1, 1, 1)
tap = torch.arange(9, device=dev).view(1, -1, 1)
grp = torch.arange(16, device=dev).view(1, 1, -1)
idx = (pix * 432 + grp * 18 + 2 * tap).reshape(-1)
flat = om.view(-1)
gout = torch.empty(idx.numel(), device=dev)
fn = lambda: torch.index_select(flat, 0, idx, out=gout)
fn()
torch.cuda.synchronize()
r = gout.clone()
nbad = 0
for it in range(12):
    gout.fill_(0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(4):
            offender()
    fn()
    torch.cuda.synchronize()
    nbad += int(not torch.equal(gout, r))
print(f"victim torch.index_select with deform_cols's om access pattern: wrong in {nbad} of 12 runs", flush=True)

# ---- neighbours that are NOT ours: rocBLAS / hipBLASLt GEMMs (MFMA, many registers), a big torch conv
ma, mb = torch.randn(8192, 4096, device=dev).half(), torch.randn(4096, 8192, device=dev).half()
mc = torch.empty(8192, 8192, device=dev, dtype=torch.float16)
fa, fb = torch.randn(4096, 2048, device=dev), torch.randn(2048, 4096, device=dev)
fc = torch.empty(4096, 4096, device=dev)
for name, fn in (("torch.mm f16 8192x4096x8192", lambda: torch.mm(ma, mb, out=mc)), ("torch.mm f32 4096x2048x4096", lambda: torch.mm(fa, fb, out=fc))):
    fn()
    torch.cuda.synchronize()
    nbad, cnt = 0, []
    for it in range(12):
        cols.fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):
                fn()
        ops.deform_cols(prop, None, om, cols, flow=flow)
        torch.cuda.synchronize()
        d = cols != ref
        if bool(d.any()):
            nbad += 1
            cnt.append(int(d.sum()))
    print(f"neighbour {name:45s}: deform_cols wrong in {nbad:2d} of 12 runs {cnt[:4]}", flush=True)
