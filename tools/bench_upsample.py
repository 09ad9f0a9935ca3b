"""r06: pp_upsample2x at the decoder's shapes: time and effective HBM bandwidth (input once + output once)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
for n, h, w, c in ((11, 90, 160, 128), (11, 180, 320, 64), (6, 90, 160, 128), (6, 180, 320, 64), (80, 45, 80, 128)):
    x = torch.randn(n, h, w, c, device=dev).half()
    out = torch.empty(n, 2 * h, 2 * w, c, device=dev, dtype=torch.float16)
    ref = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    for _ in range(3):
        ops.upsample2x(x, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.upsample2x(x, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = (x.numel() + out.numel()) * 2
    print(f"[{n},{h},{w},{c}] -> x2: {ms * 1e3:7.1f} us, {nbytes / ms / 1e9:5.2f} TB/s (in + out once), max |err| vs torch {float((out.float() - ref).abs().max()):.2e}, "
          f"checksum {float(out.float().sum()):.6e}", flush=True)
