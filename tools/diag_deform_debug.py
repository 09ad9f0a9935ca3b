"""r06 diagnostic: WHICH of deform_cols's inputs does a thread read wrongly next to a convolution on another stream?
Needs the debug build (bash tools/build_variant.sh deformdbg sample_kernels -DPP_DEFORM_DEBUG): every thread also writes the offsets /
mask / sample position it computed into a buffer passed through the unused x1 pointer."""
import os
import shutil
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
LIB = ROOT / "comfyui_propainter_nodes_amd" / "libpropainter_mi355.so"
shutil.copy(LIB, "/tmp/product.so")
shutil.copy(ROOT / "tools" / "variants" / "deformdbg.so", LIB)
try:
    from comfyui_propainter_nodes_amd import lib, ops
    L = lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    nw, h, w = 7, 90, 160
    prop = torch.randn(nw, h, w, 128, generator=g).half().to(dev)
    om = torch.cat([torch.randn(nw, h, w, 288, generator=g) * 3, torch.rand(nw, h, w, 144, generator=g)], 3).to(dev)
    flow = (torch.randn(nw, h, w, 2, generator=g) * 2).to(dev)
    cols = torch.empty(nw, h, w, 9 * 128, device=dev, dtype=torch.float16)
    nthreads = nw * h * w * 9 * 16
    npx = nw * h * w
    dbg = torch.zeros(nthreads, 8, device=dev)
    x16 = torch.randn(nw, h, w, 128, generator=g).half().to(dev)
    sp = ops.make_conv_spec(torch.randn(432, 128, 3, 3, generator=g) * 0.03, torch.randn(432, generator=g), torch.float16, padding=1).to(dev)
    out = torch.empty(nw, h, w, 432, device=dev)
    P, cin = ops._deform_cols_params(prop, None, om, 16, flow)
    P.cols = cols.data_ptr()
    P.x1, P.x1_C = dbg.data_ptr(), 0

    def run():
        L.call("pp_deform_cols", ops.stream_handle(cols), P)

    run()
    torch.cuda.synchronize()
    rcols, rdbg = cols.clone(), dbg.clone()
    side = torch.cuda.Stream(dev)
    for it in range(3):
        cols.fill_(0)
        dbg.fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4):
                ops.conv2d(sp, [x16], out, act="tanh")
        run()
        torch.cuda.synchronize()
        bad_c = (cols != rcols).view(-1, 8).any(1)          # per thread (8 channels)
        bad_d = (dbg != rdbg)
        names = ["dy", "dx", "m", "py", "px", "y0", "x0", "tap"]
        print(f"run {it}: threads with wrong outputs {int(bad_c.sum())}; threads whose recorded inputs differ: "
              + ", ".join(f"{n} {int(bad_d[:, i].sum())}" for i, n in enumerate(names)), flush=True)
        both = bad_c & bad_d.any(1)
        print(f"   wrong outputs WITH a wrong recorded input: {int(both.sum())}; wrong outputs with all recorded inputs right: {int((bad_c & ~bad_d.any(1)).sum())}")
        # hypothesis: dy = om_y + flow.y is formed by `v_pk_add_f32 v[2:3], v[2:3], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]` (the (dy, dx) pair
        # of `om` + the SWAPPED (x, y) pair of the flow).  If the low half's operand select is dropped for one 16-lane pass, dy gets
        # flow.x instead of flow.y: wrong - right == flow.x - flow.y
        wrong = bad_d[:, 0].nonzero().view(-1)
        pixw = wrong // 144
        fl = flow.view(-1, 2)[pixw]
        delta = dbg[wrong, 0] - rdbg[wrong, 0]
        print(f"   dy(wrong) - dy(solo) == flow.x - flow.y on {int(((delta - (fl[:, 0] - fl[:, 1])).abs() < 1e-5).sum())} of {wrong.numel()} wrong threads; "
              f"wrong threads per wave position (lane // 16): {torch.bincount((wrong % 64) // 16, minlength=4).tolist()}")
        omv = om.view(-1, 432)
        for t in wrong[:: max(1, wrong.numel() // 6)][:6].tolist():
            px_, rem = divmod(t, 144)
            tp, gg = divmod(rem, 16)
            o = omv[px_]
            print(f"      thread {t} (lane {t % 64}, pixel {px_}, tap {tp}, group {gg}): dy wrong {float(dbg[t, 0]):+.5f} solo {float(rdbg[t, 0]):+.5f} delta {float(dbg[t, 0] - rdbg[t, 0]):+.5f}; "
                  f"om_y {float(o[gg * 18 + 2 * tp]):+.5f} om_x {float(o[gg * 18 + 2 * tp + 1]):+.5f} flow.x {float(flow.view(-1, 2)[px_, 0]):+.5f} flow.y {float(flow.view(-1, 2)[px_, 1]):+.5f} "
                  f"flow of pixel+1 {flow.view(-1, 2)[min(px_ + 1, npx - 1)].tolist()} pixel-1 {flow.view(-1, 2)[px_ - 1].tolist()}; lane-16's om_y {float(omv[(t - 16) // 144][((t - 16) % 144 % 16) * 18 + 2 * ((t - 16) % 144 // 16)]):+.5f}")
        i = int(bad_c.nonzero()[0]) if bool(bad_c.any()) else None
        if i is not None:
            print(f"   first wrong thread {i}: recorded {dbg[i].tolist()} solo {rdbg[i].tolist()}; outputs {cols.view(-1, 8)[i].tolist()} solo {rcols.view(-1, 8)[i].tolist()}")
finally:
    shutil.copy("/tmp/product.so", LIB)
