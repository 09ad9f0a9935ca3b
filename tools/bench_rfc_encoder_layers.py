"""r06: the 128 -> 128 layers of flow completion's encoder at the clip's size (79 flows x 2 directions of 45 x 80): the 3x3 convolutions
with dilation 1 / 2 / 3 (mid_dilation) and the temporal (3,1,1) convolution seen as a [T] x [B*h*w] image -- time, TFLOP/s, kernel family."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["PP_CONV_TRACE"] = "1"
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
T, B, h, w, C = 79, 2, 45, 80, 128


def run(name, spec, x, out):
    print(f"--- {name}", flush=True)
    ops.conv2d(spec, [x], out, act="leaky", act_param=0.2)     # (prints the family once)
    os.environ.pop("PP_CONV_TRACE", None)
    lib.reload_options()
    for _ in range(3):
        ops.conv2d(spec, [x], out, act="leaky", act_param=0.2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d(spec, [x], out, act="leaky", act_param=0.2)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * out.numel() / spec.cout * spec.cout * C * spec.kh * spec.kw
    print(f"    {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.0f} TF/s", flush=True)
    os.environ["PP_CONV_TRACE"] = "1"
    lib.reload_options()


x = torch.randn(T * B, h, w, C, device=dev).half()
out = torch.empty_like(x)
for d in (1, 2, 3):
    spec = ops.make_conv_spec(torch.randn(C, C, 3, 3) * 0.03, torch.randn(C), torch.float16, padding=d, dilation=d).to(dev)
    run(f"3x3 dilation {d}", spec, x, out)
spec = ops.make_conv_spec(torch.randn(C, C, 3, 1) * 0.05, torch.randn(C), torch.float16, padding=(1, 0)).to(dev)
run("temporal (3,1) on [1, T, B*h*w, C]", spec, x.view(1, T, B * h * w, C), out.view(1, T, B * h * w, C))
