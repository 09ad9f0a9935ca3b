"""r06: RAFT's convc1 (1x1, 324 -> 256 on the lookup's output, PP_F32X2) and the all-pairs volume GEMM under the flat kernel's tile
families (PP_CONV_TILE): time and algorithmic TFLOP/s (x3 products on the matrix pipe)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
P, h, w = 158, 45, 80
corr = torch.randn(P, h, w, 352, device=dev)[..., :324]
spec = ops.make_conv_spec(torch.randn(256, 324, 1, 1) * 0.05, torch.randn(256), torch.float32, split=True).to(dev)
out = torch.empty(P, h, w, 256, device=dev)
ref = None
for tile in ("", "large", "", "large", "", "large", "xlforce", "classic"):
    if tile:
        os.environ["PP_CONV_TILE"] = tile
    else:
        os.environ.pop("PP_CONV_TILE", None)
    lib.reload_options()
    ops._PARAMS.clear()
    for _ in range(10):
        ops.conv2d(spec, [corr], out, act="relu")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d(spec, [corr], out, act="relu")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    if ref is None:
        ref = out.clone()
    print(f"convc1 PP_CONV_TILE={tile or 'default':8s}: {ms * 1e3:7.1f} us, {2.0 * P * h * w * 324 * 256 / ms / 1e9:5.0f} TF/s algorithmic, "
          f"{(corr.numel() + out.numel()) * 4 / ms / 1e9:5.2f} TB/s in + out, equal to default: {bool(torch.equal(ref, out))}", flush=True)
