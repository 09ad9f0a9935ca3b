"""r06: flow completion's decoder layers with 32 output channels at the clip's size (158 images of 180 x 320 / 360 x 640): flat 32-channel
tiles (default) against the halo-tile kernel on 64-channel tiles (PP_CONV_HALO_MINCOUT=17)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
for name, n, h, w, cin, cout in (("dec1_2 64->32", 158, 180, 320, 64, 32), ("up_0 32->32", 158, 180, 320, 32, 32), ("gen dec 64->64 (control)", 11, 360, 640, 64, 64)):
    x = torch.randn(n, h, w, cin, device=dev).half()
    spec = ops.make_conv_spec(torch.randn(cout, cin, 3, 3) * 0.05, torch.randn(cout), torch.float16, padding=1, many_images=True).to(dev)
    outs, line = [], f"{name:26s} M {n * h * w}"
    for rep in range(2):
        for mc in ("33", "17"):
            os.environ["PP_CONV_HALO_MINCOUT"] = mc
            lib.reload_options()
            ops._PARAMS.clear()
            out = torch.empty(n, h, w, cout, device=dev, dtype=torch.float16)
            for _ in range(3):
                ops.conv2d(spec, [x], out, act="leaky", act_param=0.2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv2d(spec, [x], out, act="leaky", act_param=0.2)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            outs.append(out.clone())
            line += f" | min_cout {mc}: {ms * 1e3:7.1f} us {(x.numel() + out.numel()) * 2 / ms / 1e9:5.2f} TB/s"
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    print(line + f" | max |diff| {d:.2e}", flush=True)
