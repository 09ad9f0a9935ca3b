"""Micro-benchmark of pp_conv2d on the MI355X at the hot-path shapes (run via gpurun)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

SHAPES = [
    # name, dtype, N, H, W, segC, Cout, k, stride, pad
    ("raft_gru_1x5_f32", torch.float32, 32, 45, 80, [128, 256], 128, (1, 5), 1, (0, 2)),
    ("raft_convc2_f32", torch.float32, 32, 45, 80, [256], 192, 3, 1, 1),
    ("raft_fnet_l1_f32", torch.float32, 8, 180, 320, [64], 64, 3, 1, 1),
    ("raft_gru_1x5_f32x2", "f32x2", 32, 45, 80, [128, 256], 128, (1, 5), 1, (0, 2)),
    ("raft_convc2_f32x2", "f32x2", 32, 45, 80, [256], 192, 3, 1, 1),
    ("raft_fnet_l1_f32x2", "f32x2", 8, 180, 320, [64], 64, 3, 1, 1),
    ("raft_fh2_f32x2", "f32x2", 32, 45, 80, [256], 2, 3, 1, 1),
    ("enc_conv_256_384_f16", torch.float16, 8, 90, 160, [256], 384, 3, 1, 1),
    ("dcn_offset_f16", torch.float16, 8, 90, 160, [128, 128, 8], 128, 3, 1, 1),
    ("fc1_f16", torch.float16, 1, 1, 29160, [512], 1960, 1, 1, 0),
    ("dec_64_3_f16", torch.float16, 4, 360, 640, [64], 3, 3, 1, 1),
]


def main():
    lib.load()
    dev = torch.device("cuda:0")
    res = []
    for name, dt, N, H, W, segC, Cout, k, s, p in SHAPES:
        kk = (k, k) if isinstance(k, int) else k
        split = dt == "f32x2"
        dt = torch.float32 if split else dt
        x = [torch.randn(N, H, W, c, device=dev).to(dt) for c in segC]
        w = torch.randn(Cout, sum(segC), *kk) * 0.05
        spec = ops.make_conv_spec(w, torch.zeros(Cout), dt, stride=s, padding=p, seg_channels=segC, split=split).to(dev)
        ho, wo = spec.out_hw(H, W)
        out = torch.empty(N, ho, wo, Cout, device=dev, dtype=dt)
        for _ in range(3):
            ops.conv2d(spec, x, out, act="relu")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for _ in range(iters):
            ops.conv2d(spec, x, out, act="relu")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        flops = 2.0 * N * ho * wo * Cout * sum(segC) * kk[0] * kk[1]
        res.append({"name": name, "ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 2)})
        print(res[-1], flush=True)
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/bench_conv.json").write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
