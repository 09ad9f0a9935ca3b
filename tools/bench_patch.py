"""r06: the patch-gathering PP_F32X2 convolution (conv_patch.hip) against pp_im2col + 1x1 at RAFT's real shapes (MI355X)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

CASES = [("convf1 7x7 2->128 on the flow view, 158x45x80", 158, 45, 80, 2, 128, 7, 1, 3, True),
         ("stem 7x7/2 3->64 on 16 frames of 360x640", 16, 360, 640, 3, 64, 7, 2, 3, False)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib.load()
    dev = torch.device("cuda:0")
    for name, n, h, w, c, cout, k, s, p, view in CASES:
        kv = k * k * c
        kpad = ops.pad32(kv)
        wt = torch.randn(cout, kv, 1, 1) * 0.1
        spec = ops.make_conv_spec(wt, torch.randn(cout), torch.float32, seg_channels=[kpad], seg_valid=[kv], split=True).to(dev)
        if view:
            x = torch.randn(n, h, w, 128, device=dev)[..., 126:128]
        else:
            x = torch.randn(n, h, w, c, device=dev)
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        out = torch.empty(n, ho, wo, cout, device=dev)
        cols = torch.empty(n, ho, wo, kpad, device=dev)
        t_patch = timed(lambda: ops.conv2d_patch(spec, x, out, k, k, stride=s, padding=p, act="relu"))
        t_i = timed(lambda: ops.im2col(x, cols, k, k, stride=s, padding=p))
        t_g = timed(lambda: ops.conv2d(spec, [cols], out, act="relu"))
        gf = 2.0 * n * ho * wo * cout * kv
        print(f"{name}: patch kernel {t_patch:.1f} us ({gf / t_patch / 1e6:.0f} TF/s, output {out.numel() * 4 / t_patch / 1e6:.2f} TB/s) | "
              f"im2col {t_i:.1f} + 1x1 {t_g:.1f} = {t_i + t_g:.1f} us")


if __name__ == "__main__":
    main()
