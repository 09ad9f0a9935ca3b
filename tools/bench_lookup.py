"""r06: pp_corr_lookup_conv (lookup fused with the 324 -> 256 projection) against pp_corr_lookup + 1x1 PP_F32X2 at RAFT's update-block
size (158 pair-directions x 45 x 80 pixels, tiled levels 0 / 1) on the MI355X."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
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
    P, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 158, 45, 80
    hw = h * w
    pyr = []
    hi, wi = h, w
    for lvl in range(4):
        tiled = lvl < 2
        pitch = ops.tiled_pitch(hi, wi) if tiled else hi * wi
        pyr.append((torch.randn(P, hw, pitch, device=dev), hi, wi, tiled))
        hi, wi = hi // 2, wi // 2
    mf = torch.randn(P, h, w, 128, device=dev) * 3
    flow = mf[..., 126:128]
    spec = ops.make_conv_spec(torch.randn(256, 324, 1, 1) * 0.05, torch.randn(256), torch.float32, split=True).to(dev)
    corr = torch.empty(P, h, w, 352, device=dev)[..., :324]
    two, one = torch.empty(P, h, w, 256, device=dev), torch.empty(P, h, w, 256, device=dev)
    t_l = timed(lambda: ops.corr_lookup(pyr, flow, corr))
    t_c = timed(lambda: ops.conv2d(spec, [corr], two, act="relu"))
    t_f = timed(lambda: ops.corr_lookup_conv(pyr, flow, spec, one, act="relu"))
    d = (one - two).abs().max().item()
    npx = P * hw
    print(f"{P} x {h} x {w}: lookup {t_l:.1f} us + convc1 {t_c:.1f} us = {t_l + t_c:.1f} us | fused {t_f:.1f} us "
          f"({npx * (4 * 400 + 1024) / t_f / 1e6:.2f} TB/s of the lookup's bytes without its output, {2.0 * npx * 256 * 324 / t_f / 1e6:.0f} TF/s) | max |diff| {d:.2e}")


if __name__ == "__main__":
    main()
