"""How much the flow-completion stage amplifies a 1.4e-4 px perturbation of its input at the 80-frame 640x360 bench clip, per
synthetic-weight variant (MI355X; diagnostic for the choice of the `contractive` variant, weights._synth_tensor)."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from comfyui_propainter_nodes_amd import lib, pipeline, weights  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
C = dict(bench.CFG)
if len(sys.argv) > 1:          # WxHxT, e.g. 1280x720x90 (BASELINE configs[4]'s size, two sub-videos)
    C["W"], C["H"], C["T"] = (int(v) for v in sys.argv[1].split("x"))
VARIANTS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["", "contractive", "undamped"]
frames_u8, fm, md = bench.make_inputs(C["T"], C["H"], C["W"], C["mask_dilates"], C["flow_mask_dilates"])
fr, fmd = torch.from_numpy(frames_u8).to(dev), torch.from_numpy(fm).to(dev)
for variant in VARIANTS:
    sds = weights.synth_state_dicts(0, variant)
    for fp16 in ("disable", "enable"):
        models = pipeline.models_from_state_dicts(sds, dev, fp16)
        cfg = pipeline.ProPainterConfig(C["ref_stride"], C["neighbor_length"], C["subvideo_length"], C["raft_iter"], fp16, C["T"], dev, (C["W"], C["H"]))
        from comfyui_propainter_nodes_amd import ops
        gt = pipeline.compute_flow(models.raft_model, ops.frames_from_u8(fr), cfg)
        a = pipeline.complete_flow(models.flow_model, gt, fmd, cfg.subvideo_length).clone()
        g = torch.Generator(device=dev).manual_seed(7)
        gt2 = gt + 1.4e-4 * torch.randn(gt.shape, device=dev, generator=g)
        if fp16 == "disable" and os.environ.get("PP_DIAG_CROSS") == "1":   # fp32-storage stage against the f16-storage stage, same input
            other = pipeline.models_from_state_dicts(sds, dev, "enable")
            c = pipeline.complete_flow(other.flow_model, gt, fmd, cfg.subvideo_length)
            dd = (a - c).abs()
            print(f"   fp32-storage vs f16-storage stage on identical RAFT flows: max {float(dd.max()):.3e} mean {float(dd.mean()):.3e}", flush=True)
            del other, c
        b = pipeline.complete_flow(models.flow_model, gt2, fmd, cfg.subvideo_length)
        d = (a - b).abs()
        print(f"variant {variant or 'default':12s} fp16 {fp16:8s}: completed flow absmax {float(a.abs().max()):9.2f}  finite {bool(torch.isfinite(a).all())}  "
              f"response to a 1.4e-4 px input perturbation: max {float(d.max()):.3e} mean {float(d.mean()):.3e}", flush=True)
        del models
