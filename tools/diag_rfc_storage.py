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

# ---- inside the decoder (the stage that parted above): every intermediate of FlowCompleter._decode in both storage types
if len(sys.argv) > 3 and sys.argv[3] == "decode":
    def stages(C, prop, e1):
        dev = prop.device
        n, h, w, _ = prop.shape
        new = lambda hh, ww, c: torch.empty(n, hh, ww, c, device=dev, dtype=C.dt)
        out = {}
        out["a"] = ops.conv2d(C.dec2_0, [prop], new(h, w, 128), act="leaky", act_param=0.2)
        out["up1"] = ops.upsample2x(out["a"], new(2 * h, 2 * w, 128))
        out["b"] = ops.conv2d(C.dec2_2, [out["up1"]], new(2 * h, 2 * w, 64), act="leaky", act_param=0.2, epi="add", aux1=e1)
        out["c"] = ops.conv2d(C.dec1_0, [out["b"]], new(2 * h, 2 * w, 64), act="leaky", act_param=0.2)
        out["up2"] = ops.upsample2x(out["c"], new(4 * h, 4 * w, 64))
        out["d"] = ops.conv2d(C.dec1_2, [out["up2"]], new(4 * h, 4 * w, 32), act="leaky", act_param=0.2)
        out["e"] = ops.conv2d(C.up_0, [out["d"]], new(4 * h, 4 * w, 32), act="leaky", act_param=0.2)
        out["up3"] = ops.upsample2x(out["e"], new(8 * h, 8 * w, 32))
        out["pred"] = ops.conv2d(C.up_2, [out["up3"]], new(8 * h, 8 * w, 2))
        return out
    import os
    os.environ["PP_CONV_TRACE"] = "1"
    lib.reload_options()
    x16 = torch.empty(T - 1, 2, H, W, 4, device=dev, dtype=torch.float16); ops.rfc_prep(gt, fmd, x16)
    x32 = torch.empty(T - 1, 2, H, W, 4, device=dev, dtype=torch.float32); ops.rfc_prep(gt, fmd, x32)
    e16, _ = m16.flow_model._encode(x16)
    e32, _ = m32.flow_model._encode(x32)
    s16 = stages(m16.flow_model, t16["prop"], e16)
    print("---- f32 storage launches:", flush=True)
    s32 = stages(m32.flow_model, t32["prop"], e32)
    for k in s16:
        d = (s16[k].float() - s32[k].float()).abs()
        per = d.flatten(1).max(1).values
        bad = torch.nonzero(per > 1.0).flatten().tolist()
        print(f"{k:5s} {tuple(s16[k].shape)} diff max {float(d.max()):.3e}  first bad image {bad[0] if bad else None} of {len(per)}", flush=True)
