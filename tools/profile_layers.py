"""Per-layer conv timing of one full pass (HIP events around every pp_conv2d launch)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from comfyui_propainter_nodes_amd import lib, ops, pipeline, weights  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
sds, _ = weights.get_state_dicts(0)
models = pipeline.models_from_state_dicts(sds, dev)
C = bench.CFG
frames_u8, fm, md = bench.make_inputs(C["T"], C["H"], C["W"], C["mask_dilates"], C["flow_mask_dilates"])
cfg = pipeline.ProPainterConfig(C["ref_stride"], C["neighbor_length"], C["subvideo_length"], C["raft_iter"], "enable", C["T"], dev,
                                (C["W"], C["H"]))
args = (models, torch.from_numpy(frames_u8).to(dev), torch.from_numpy(fm).to(dev), torch.from_numpy(md).to(dev), cfg)
pipeline.run_inpainting(*args, to_host=False)
ops.CONV_PROFILE = ops.ConvProfile(detailed=True)
pipeline.run_inpainting(*args, to_host=False)
prof = ops.CONV_PROFILE.summary()
ops.CONV_PROFILE = None
tot = sum(v["ms"] for v in prof.values())
print(f"total conv ms {tot:.1f}")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:70]:
    print(f"{v['ms']:8.1f} ms  {v['n']:5d}x  {v['flops'] / v['ms'] / 1e9:7.1f} TF/s  {k}")
