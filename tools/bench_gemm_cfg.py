"""r06 A/B: tile configurations of conv_gemm_f16_kernel (PP_CONV_GEMM_CFG) at the transformer's shapes, with the paired-quad stores."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
SHAPES = (("ss (materialised) K 6272", 17820, 6272, 512, False), ("kv pooled tokens", 1547, 512, 1024, False), ("qkv", 30780, 512, 1536, False))
for name, M, K, Cout, residual in SHAPES + (("qkv", 30780, 512, 1536, False), ("proj + residual", 30780, 512, 512, True), ("fc1", 30780, 512, 1960, False),
                                   ("fc2 (materialised) + residual", 30780, 1960, 512, True), ("sc fc", 17820, 512, 6272, False),
                                   ("dcn 1x1", 201600, 1152, 128, False), ("qkv 6-frame window", 9720, 512, 1536, False))[:0]:
    x = torch.randn(1, 1, M, K, device=dev).half()
    spec = ops.make_conv_spec(torch.randn(Cout, K, 1, 1) * 0.05, torch.randn(Cout), torch.float16).to(dev)
    res = torch.randn(1, 1, M, Cout, device=dev).half()
    kw = dict(epi="add", aux1=res) if residual else {}
    outs, line = [], f"{name:30s}"
    for rep in range(2):
        for cfg in ("0", "7", "5", "6"):
            os.environ["PP_CONV_GEMM_CFG"] = cfg
            os.environ["PP_CONV_GEMM"] = "force"
            lib.reload_options()
            ops._PARAMS.clear()
            out = torch.empty(1, 1, M, Cout, device=dev, dtype=torch.float16)
            for _ in range(3):
                ops.conv2d(spec, [x], out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.conv2d(spec, [x], out, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            outs.append(out.clone())
            line += f" | cfg {cfg}: {ms * 1e3:6.1f} us {2.0 * M * K * Cout / ms / 1e9:4.0f}"
    print(line + f" | equal: {all(torch.equal(outs[0], o) for o in outs[1:])}", flush=True)
