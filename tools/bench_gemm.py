"""Micro-benchmark of the 1x1 f16 GEMM layers (the transformer's Linear layers, the deformable 1x1, the soft-composite fc) on the
MI355X: the flat implicit-GEMM kernel against every tile configuration of conv_gemm_f16.hip (run via gpurun)."""
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

SHAPES = [  # name, M, K, Cout, activation
    ("qkv 512->1536", 27540, 512, 1536, None),
    ("proj 512->512", 27540, 512, 512, None),
    ("fc1 512->1960", 27540, 512, 1960, None),
    ("fc2 1960->512", 27540, 1960, 512, None),
    ("fc 512->6272", 17820, 512, 6272, None),
    ("dcn 1152->128", 201600, 1152, 128, None),
]


def main():
    lib.load()
    dev = torch.device("cuda:0")
    res = []
    cfgs = [("flat", dict(PP_CONV_GEMM="0"))] + [(f"gemm{c}", dict(PP_CONV_GEMM="force", PP_CONV_GEMM_CFG=str(c))) for c in (4, 5, 6)]
    for name, M, K, Cout, act in SHAPES:
        x = torch.randn(1, 1, M, K, device=dev).half()
        w = torch.randn(Cout, K, 1, 1) * 0.05
        spec = ops.make_conv_spec(w, torch.zeros(Cout), torch.float16).to(dev)
        out = torch.empty(1, 1, M, Cout, device=dev, dtype=torch.float16)
        row = {"name": name}
        ref = None
        for cname, env in cfgs:
            os.environ.update(env)
            lib.reload_options()
            for _ in range(3):
                ops.conv2d(spec, [x], out)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(ref, out))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 30
            e0.record()
            for _ in range(iters):
                ops.conv2d(spec, [x], out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            row[cname] = f"{ms * 1e3:.1f} us {2.0 * M * K * Cout / ms / 1e9:.0f} TF/s" + ("" if same else " MISMATCH")
        res.append(row)
        print(row, flush=True)
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/bench_gemm.json").write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
