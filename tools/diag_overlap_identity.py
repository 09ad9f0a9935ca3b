"""Is flows_overlapped() bit-identical to compute_flow() + complete_flow()?  (MI355X; diagnostic)"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from comfyui_propainter_nodes_amd import lib, ops, pipeline, weights  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
W, H, T = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "640x360x200").split("x"))
frames_u8, fm, md = bench.make_inputs(T, H, W, 5, 8)
fr, fmd = torch.from_numpy(frames_u8).to(dev), torch.from_numpy(fm).to(dev)
models = pipeline.models_from_state_dicts(weights.synth_state_dicts(0), dev, "enable")
cfg = pipeline.ProPainterConfig(10, 10, 80, 20, "enable", T, dev, (W, H))
frames = ops.frames_from_u8(fr)
for rep in range(2):
    gt0 = pipeline.compute_flow(models.raft_model, frames, cfg).clone()
    pr0 = pipeline.complete_flow(models.flow_model, gt0, fmd, 80).clone()
    gt1, pr1 = pipeline.flows_overlapped(models, frames, fmd, cfg)
    torch.cuda.synchronize()
    dg, dp = (gt0 - gt1).abs(), (pr0 - pr1).abs()
    print(f"rep {rep}: RAFT flows identical {bool(torch.equal(gt0, gt1))} (max {float(dg.max()):.3e}, first differing pair "
          f"{[int(v) for v in torch.nonzero(dg.flatten(2).max(2).values.max(0).values > 0).flatten()[:6].tolist()]}); completed identical {bool(torch.equal(pr0, pr1))} (max {float(dp.max()):.3e})", flush=True)
    # serial completion on the overlapped RAFT flows: is the completion itself stable under concurrency?
    pr2 = pipeline.complete_flow(models.flow_model, gt1, fmd, 80)
    print(f"        completion(serial, on the overlapped RAFT flows) == overlapped completion: {bool(torch.equal(pr2, pr1))}", flush=True)
