"""Per-frame profile of the completed-flow difference against a node fixture (MI355X; diagnostic)."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("PP_ALLOW_SYNTHETIC_WEIGHTS", "1")
from comfyui_propainter_nodes_amd import nodes, synth  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "cfg2_80f_node"
fp16 = sys.argv[2] if len(sys.argv) > 2 else "disable"
g = np.load(ROOT / "tests" / "golden" / f"{case}.npz")
P = json.loads(str(g["params_json"]))
image, mask = synth.synthetic_clip(P["T"], P["H"], P["W"])
if P.get("mask_kind", "static") == "moving":
    mask = synth.moving_mask(P["T"], P["H"], P["W"])
common = {k: P[k] for k in ("mask_dilates", "flow_mask_dilates", "ref_stride", "neighbor_length", "subvideo_length", "raft_iter")}
nodes.TRACE = tr = {}
nodes.ProPainterInpaint().propainter_inpainting(image, mask, P["width"], P["height"], fp16=fp16, **common)
s = P["flow_stride"]
pf = tr["pred_flows"].cpu()[:, :, ::s, ::s].permute(0, 1, 4, 2, 3).numpy()
d = np.abs(pf - g["pred_flow"].astype(np.float32))        # [2, T-1, 2, h/s, w/s]
gt = tr["gt_flows"].cpu()[:, :, ::2 * s, ::2 * s].permute(0, 1, 4, 2, 3).numpy()
dg = np.abs(gt - g["gt_flow"])
print(case, fp16, "pred max", d.max(), "mean", d.mean(), " gt max", dg.max())
for dirn in (0, 1):
    print("dir", dirn, "per-frame max:", " ".join(f"{v:.2g}" for v in d[dirn].max(axis=(1, 2, 3))))
    print("dir", dirn, "per-frame gt max:", " ".join(f"{v:.1g}" for v in dg[dirn].max(axis=(1, 2, 3))))
big = np.argwhere(d > 0.5)
print("entries > 0.5 px:", len(big), "of", d.size, "; y range", big[:, 3].min() * s if len(big) else None, big[:, 3].max() * s if len(big) else None,
      "x range", big[:, 4].min() * s if len(big) else None, big[:, 4].max() * s if len(big) else None)

# ---- teacher-forced: the CPU oracle's flow completion on OUR RAFT flows (same input on both sides) -------------------
if os.environ.get("PP_DIAG_ORACLE", "1") == "1":
    import time
    from comfyui_propainter_nodes_amd import weights
    from oracle import pipeline as OP

    sd = weights.synth_state_dicts(P["seed"])["rfc"]
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    gtf = tr["gt_flows"].cpu()                                      # [2,T-1,H,W,2]
    fl = (gtf[0].permute(0, 3, 1, 2)[None].contiguous(), gtf[1].permute(0, 3, 1, 2)[None].contiguous())
    fm = tr["flow_masks"].cpu().float()[None, :, None]
    t0 = time.time()
    with torch.no_grad():
        ref = OP.complete_flow(sd, fl, fm, P["subvideo_length"])
    print(f"oracle flow completion on our RAFT flows: {time.time() - t0:.1f} s")
    ours = tr["pred_flows"].cpu()
    for dirn in (0, 1):
        dd = (ours[dirn].permute(0, 3, 1, 2) - ref[dirn][0]).abs()
        print("teacher-forced dir", dirn, "max", dd.max().item(), "mean", dd.mean().item(), "per-frame max:",
              " ".join(f"{v:.2g}" for v in dd.amax(dim=(1, 2, 3)).tolist()))
    rf = torch.stack([ref[0][0], ref[1][0]], 0)[:, :, :, ::s, ::s].numpy()
    ds = np.abs(rf - g["pred_flow"].astype(np.float32))
    print("oracle(our RAFT flows) vs reference fixture (the reference's own sensitivity to our 1.4e-4 px input difference): max",
          ds.max(), "mean", ds.mean())
