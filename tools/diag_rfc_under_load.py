"""Does flow completion give the same bits while another stream keeps the chip busy?  (MI355X; diagnostic, r04: the sub-video
overlap exposed a difference)  Runs FlowCompleter on fixed flows (a) alone, (b) while RAFT runs on another stream, and compares
the traced stage tensors bit for bit."""
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
W, H, T = 640, 360, 86
frames_u8, fm, md = bench.make_inputs(T, H, W, 5, 8)
fr, fmd = torch.from_numpy(frames_u8).to(dev), torch.from_numpy(fm).to(dev)
models = pipeline.models_from_state_dicts(weights.synth_state_dicts(0), dev, "enable")
cfg = pipeline.ProPainterConfig(10, 10, 80, 20, "enable", T, dev, (W, H))
frames = ops.frames_from_u8(fr)
gt = pipeline.compute_flow(models.raft_model, frames, cfg)
torch.cuda.synchronize()
C = models.flow_model
side = torch.cuda.Stream(dev)


def run(load: bool, traced: bool):
    tr = {} if traced else None
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        out = C(gt, fmd, trace=tr)
    if load:
        pipeline.compute_flow(models.raft_model, frames[:40], cfg)      # ~110 ms of large kernels on the launch stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return out.clone(), ({k: v.clone() for k, v in tr.items()} if traced else {})


for traced in (True, False):
    base, btr = run(False, traced)
    again, _ = run(False, traced)
    print(f"traced(eager recurrence)={traced}: quiet run repeatable {bool(torch.equal(base, again))}", flush=True)
    for rep in range(3):
        out, tr = run(True, traced)
        msg = f"   under load, rep {rep}: output identical {bool(torch.equal(out, base))} (max diff {float((out - base).abs().max()):.3e})"
        for k in ("mid", "prop", "pred"):
            if k in tr:
                msg += f"; {k} identical {bool(torch.equal(tr[k], btr[k]))}"
        print(msg, flush=True)
