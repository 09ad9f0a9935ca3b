"""Where do the fp32-storage and the f16-storage flow-completion stages part?  (MI355X; diagnostic)  Runs FlowCompleter in both
storage types on the same RAFT flows at WxHxT and prints the difference of every traced tensor."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from comfyui_propainter_nodes_amd import lib, ops, pipeline, weights  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
W, H, T = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1280x720x40").split("x"))
variant = sys.argv[2] if len(sys.argv) > 2 else "contractive"
frames_u8, fm, md = bench.make_inputs(T, H, W, 5, 8)
fr, fmd = torch.from_numpy(frames_u8).to(dev), torch.from_numpy(fm).to(dev)
sds = weights.synth_state_dicts(0, variant)
m16 = pipeline.models_from_state_dicts(sds, dev, "enable")
m32 = pipeline.models_from_state_dicts(sds, dev, "disable")
cfg = pipeline.ProPainterConfig(10, 20, 80, 20, "enable", T, dev, (W, H))
gt = pipeline.compute_flow(m16.raft_model, ops.frames_from_u8(fr), cfg)
print("RAFT flows", tuple(gt.shape), "absmax", float(gt.abs().max()))
t16, t32 = {}, {}
a = m16.flow_model(gt, fmd, trace=t16)
b = m32.flow_model(gt, fmd, trace=t32)
for k in ("mid", "prop", "pred"):
    x, y = t16[k].float(), t32[k].float()
    d = (x - y).abs()
    print(f"{k:5s} {tuple(x.shape)}: f16 absmax {float(x.abs().max()):.3f} f32 absmax {float(y.abs().max()):.3f}  diff max {float(d.max()):.3e} mean {float(d.mean()):.3e}", flush=True)
    if d.max() > 1:
        idx = torch.nonzero(d > 0.5 * d.max())[0].tolist()
        print("      first large entry at", idx, "per leading index max:", [round(float(v), 3) for v in d.flatten(1).max(1).values[:12].tolist()])
d = (a - b).abs()
print("out  diff max", float(d.max()), "mean", float(d.mean()))
