"""r06 yardstick: what does the vendor GEMM (torch.nn.functional.linear -> hipBLASLt / rocBLAS) reach on the transformer's Linear shapes,
next to conv_gemm_f16_kernel?  Not used by the product; the number says how far the hand-written kernel is from the library's."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device("cuda:0")


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, M, K, N in (("qkv", 30780, 512, 1536), ("proj", 30780, 512, 512), ("fc1", 30780, 512, 1960), ("fc2", 30780, 1960, 512),
                      ("sc fc", 17820, 512, 6272), ("dcn 1x1", 201600, 1152, 128)):
    x = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K) * 0.05)
    b = torch.randn(N)
    wd, bd = w.half().to(dev), b.half().to(dev)
    spec = ops.make_conv_spec(w.reshape(N, K, 1, 1), b, torch.float16).to(dev)
    out = torch.empty(1, 1, M, N, device=dev, dtype=torch.float16)
    t_ours = timed(lambda: ops.conv2d(spec, [x.view(1, 1, M, K)], out))
    t_lib = timed(lambda: F.linear(x, wd, bd))
    res = torch.randn(M, N, device=dev).half()
    t_lib_add = timed(lambda: torch.addmm(res, x, wd.t()))
    fl = 2.0 * M * K * N
    err = float((F.linear(x, wd, bd).float() - out.view(M, N).float()).abs().max())
    print(f"{name:8s} M {M} K {K} N {N}: ours {t_ours * 1e3:6.1f} us {fl / t_ours / 1e9:5.0f} TF/s | F.linear(+bias) {t_lib * 1e3:6.1f} us {fl / t_lib / 1e9:5.0f} TF/s | "
          f"addmm(+residual) {t_lib_add * 1e3:6.1f} us | max |ours - lib| {err:.2e}", flush=True)
