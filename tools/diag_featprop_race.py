"""r06 diagnostic: which kernel of the feature-propagation step gives different bits when an identical step runs next to it on a
second stream?  (tools/diag_cfg5_repro.py: PP_FEATPROP_LANES=2 is not reproducible unless PP_DEFORM_FUSED=force.)  Fixed inputs; per
iteration the step's kernels run on the launch stream while the same kernels run on a side stream on other buffers; every output is
compared with the solo run's."""
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


def make():
    d = {}
    d["prop"] = torch.randn(nw, h, w, 128, generator=g).half().to(dev)
    d["cur"] = torch.randn(nw, h, w, 128, generator=g).half().to(dev)
    d["om"] = torch.cat([torch.randn(nw, h, w, 288, generator=g) * 3, torch.rand(nw, h, w, 144, generator=g)], 3).to(dev)
    d["flow"] = (torch.randn(nw, h, w, 2, generator=g) * 2).to(dev)
    d["cols"] = torch.empty(nw, h, w, 9 * 128, device=dev, dtype=torch.float16)
    d["aligned"] = torch.empty(nw, h, w, 128, device=dev, dtype=torch.float16)
    d["warped"] = torch.empty(nw, h, w, 128, device=dev, dtype=torch.float16)
    d["t128"] = torch.empty(nw, h, w, 128, device=dev, dtype=torch.float16)
    d["om2"] = torch.empty(nw, h, w, 432, device=dev, dtype=torch.float32)
    return d


dcn = ops.make_conv_spec(torch.randn(128, 9 * 128, 1, 1, generator=g) * 0.03, torch.randn(128, generator=g), torch.float16).to(dev)
c33 = ops.make_conv_spec(torch.randn(128, 128, 3, 3, generator=g) * 0.03, torch.randn(128, generator=g), torch.float16, padding=1).to(dev)
off6 = ops.make_conv_spec(torch.randn(432, 128, 3, 3, generator=g) * 0.03, torch.randn(432, generator=g), torch.float16, padding=1).to(dev)


def step(d):
    ops.flow_warp(d["prop"], d["flow"], d["warped"])
    ops.conv2d(c33, [d["cur"]], d["t128"], act="leaky", act_param=0.1)
    ops.conv2d(off6, [d["t128"]], d["om2"], act="tanh", out_scale=3.0, act2="sigmoid", act_split=288)
    ops.deform_cols(d["prop"], None, d["om"], d["cols"], flow=d["flow"])
    ops.conv2d(dcn, [d["cols"]], d["aligned"])


A, B = make(), make()
step(A)
torch.cuda.synchronize()
ref = {k: A[k].clone() for k in ("warped", "t128", "om2", "cols", "aligned")}
inputs = {k: A[k].clone() for k in ("prop", "cur", "om", "flow")}
side = torch.cuda.Stream(dev)
WHICH = os.environ.get("DIAG_NEIGHBOUR", "step")     # what runs on the side stream: the whole step or one of its kernels
neigh = {"step": step,
         "warp": lambda d: ops.flow_warp(d["prop"], d["flow"], d["warped"]),
         "c33": lambda d: ops.conv2d(c33, [d["cur"]], d["t128"], act="leaky", act_param=0.1),
         "off6": lambda d: ops.conv2d(off6, [d["t128"]], d["om2"], act="tanh", out_scale=3.0, act2="sigmoid", act_split=288),
         "cols": lambda d: ops.deform_cols(d["prop"], None, d["om"], d["cols"], flow=d["flow"]),
         "dcn": lambda d: ops.conv2d(dcn, [d["cols"]], d["aligned"])}
for which in WHICH.split(","):
    bad = {k: 0 for k in ref}
    binp = {k: 0 for k in inputs}
    first = None
    N = 30
    for it in range(N):
        for k in ref:
            A[k].fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4):
                neigh[which](B)
        step(A)
        torch.cuda.synchronize()
        for k in ref:
            if not torch.equal(A[k], ref[k]):
                bad[k] += 1
                if k == "cols" and first is None:
                    d = (A[k] != ref[k]).view(-1).nonzero().view(-1)
                    e = d[0].item()
                    first = (int(d.numel()), e // (9 * 128), (e % (9 * 128)) // 128, e % 128, float(A[k].view(-1)[e]), float(ref[k].view(-1)[e]),
                             int(d[-1].item()) // (9 * 128))
        for k in inputs:
            if not torch.equal(A[k], inputs[k]):
                binp[k] += 1
    print(f"neighbour = {which}: {N} iterations; outputs that differed from the solo run: {bad}; INPUTS changed: {binp}; first cols difference "
          f"(count, pixel, tap, channel, got, solo, last pixel): {first}", flush=True)

# ---- deform_cols alone, where do the differences sit?
for xcd in ("1", "0"):
    os.environ["PP_DEFORM_XCD"] = xcd
    lib.reload_options()
    ops.deform_cols(A["prop"], None, A["om"], A["cols"], flow=A["flow"])
    torch.cuda.synchronize()
    r = A["cols"].clone()
    nbad, where = 0, []
    for it in range(30):
        A["cols"].fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.deform_cols(B["prop"], None, B["om"], B["cols"], flow=B["flow"])
        ops.deform_cols(A["prop"], None, A["om"], A["cols"], flow=A["flow"])
        torch.cuda.synchronize()
        d = (A["cols"] != r)
        if bool(d.any()):
            nbad += 1
            idx = d.view(-1).nonzero().view(-1)
            if len(where) < 3:
                e = idx[0].item()
                where.append((int(idx.numel()), e // (9 * 128), (e % (9 * 128)) // 128, e % 128, float(A["cols"].view(-1)[e]), float(r.view(-1)[e])))
    print(f"deform_cols alone, PP_DEFORM_XCD={xcd}: {nbad} of 30 runs differ; (count, pixel, tap, channel, got, solo) of the first ones: {where}")

# ---- does a kernel write outside its output?  Run ONE kernel on B alone and look at every other tensor.
torch.cuda.synchronize()
step(A)
step(B)
torch.cuda.synchronize()
snap = {("A", k): v.clone() for k, v in A.items()}
snap.update({("B", k): v.clone() for k, v in B.items()})
for which, outk in (("off6", "om2"), ("dcn", "aligned"), ("c33", "t128"), ("cols", "cols")):
    for _ in range(5):
        neigh[which](B)
    torch.cuda.synchronize()
    changed = [f"{n}.{k}" for (n, k), v in snap.items() if not (n == "B" and k == outk) and not torch.equal((A if n == "A" else B)[k], v)]
    print(f"{which} on B alone: tensors other than B.{outk} that changed: {changed}", flush=True)

# ---- minimal pair: ONLY deform_cols on the launch stream, ONLY the 128 -> 432 convolution on the side stream
for xcd in ("1", "0"):
    os.environ["PP_DEFORM_XCD"] = xcd
    lib.reload_options()
    nbad, cnt = 0, []
    for it in range(30):
        A["cols"].fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4):
                neigh["off6"](B)
        neigh["cols"](A)
        torch.cuda.synchronize()
        d = A["cols"] != ref["cols"]
        if bool(d.any()):
            nbad += 1
            cnt.append(int(d.sum()))
    print(f"deform_cols(A) next to off6(B) only, PP_DEFORM_XCD={xcd}: {nbad} of 30 differ; differing elements {cnt[:6]}", flush=True)
# ... and the other way round: is the CONVOLUTION's output stable next to deform_cols?
step(B)
torch.cuda.synchronize()
refB = B["om2"].clone()
nbad = 0
for it in range(30):
    B["om2"].fill_(0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(4):
            neigh["cols"](A)
    neigh["off6"](B)
    torch.cuda.synchronize()
    nbad += int(not torch.equal(B["om2"], refB))
print(f"off6(B) next to deform_cols(A): {nbad} of 30 differ", flush=True)

# ---- which of the two results is the right one?  Recompute some differing elements on the host from the inputs.
import math
os.environ["PP_DEFORM_XCD"] = "1"
lib.reload_options()
A["cols"].fill_(0)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    for _ in range(4):
        neigh["off6"](B)
neigh["cols"](A)
torch.cuda.synchronize()
got = A["cols"].clone()
d = (got != ref["cols"]).view(-1).nonzero().view(-1)
prop, om, flow = A["prop"].float().cpu(), A["om"].cpu(), A["flow"].cpu()


def expect(pix, tap, ch):
    n, p = divmod(pix, h * w)
    y, x = divmod(p, w)
    g = ch // 8
    o = om.view(-1, 432)[pix]
    dy, dx, m = float(o[g * 18 + 2 * tap]), float(o[g * 18 + 2 * tap + 1]), float(o[288 + g * 9 + tap])
    f = flow.view(-1, 2)[pix]
    dx += float(f[0]); dy += float(f[1])
    py, px = (y - 1 + tap // 3) + dy, (x - 1 + tap % 3) + dx
    if not (py > -1 and py < h and px > -1 and px < w):
        return 0.0
    y0, x0 = math.floor(py), math.floor(px)
    ly, lx = py - y0, px - x0
    v = 0.0
    for yy, xx, wgt in ((y0, x0, (1 - ly) * (1 - lx)), (y0, x0 + 1, (1 - ly) * lx), (y0 + 1, x0, ly * (1 - lx)), (y0 + 1, x0 + 1, ly * lx)):
        if 0 <= yy < h and 0 <= xx < w:
            v += wgt * m * float(prop[n, yy, xx, ch])
    return v


print(f"{d.numel()} elements differ in this run; host recomputation of a few of them:")
for e in d[:: max(1, d.numel() // 8)][:8].tolist():
    pix, rem = divmod(e, 9 * 128)
    tap, ch = divmod(rem, 128)
    print(f"   pixel {pix} tap {tap} ch {ch}: solo {float(ref['cols'].view(-1)[e]):+.5f}  next to the convolution {float(got.view(-1)[e]):+.5f}  host {expect(pix, tap, ch):+.5f}", flush=True)
