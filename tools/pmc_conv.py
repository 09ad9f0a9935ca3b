"""One conv shape per kernel family at the real RAFT size, a few launches each: the workload for rocprofv3 --pmc
passes (tools/pmc_conv.sh)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

CASES = [
    # name, split, dtype, N, H, W, segC, Cout, k, pad
    ("gru_split", True, torch.float32, 158, 45, 80, [128, 128], 256, (1, 5), (0, 2)),
    ("gru_5x1_split", True, torch.float32, 158, 45, 80, [128, 128], 256, (5, 1), (2, 0)),
    ("convc2_split", True, torch.float32, 158, 45, 80, [256], 192, (3, 3), (1, 1)),
    ("fh1_split", True, torch.float32, 158, 45, 80, [128], 256, (3, 3), (1, 1)),
    ("enc_f16", False, torch.float16, 16, 90, 160, [256], 384, (3, 3), (1, 1)),
]


def main():
    lib.load()
    dev = torch.device("cuda:0")
    for name, split, dt, N, H, W, segC, Cout, k, p in CASES:
        x = [torch.randn(N, H, W, c, device=dev).to(dt) for c in segC]
        w = torch.randn(Cout, sum(segC), *k) * 0.05
        spec = ops.make_conv_spec(w, torch.zeros(Cout), dt, padding=p, seg_channels=segC, split=split).to(dev)
        out = torch.empty(N, *spec.out_hw(H, W), Cout, device=dev, dtype=dt)
        for _ in range(3):
            ops.conv2d(spec, x, out, act="relu")
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
