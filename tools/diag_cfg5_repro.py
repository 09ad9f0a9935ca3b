"""r06 diagnostic: is the composed output at cfg 5's geometry (1280x720, neighbor_length 20, chunked RAFT) reproducible run to run, and
the same for every lane setting?  Prints, per run, whether the uint8 frames equal the first run's and how many bytes differ."""
import os
import sys

import torch

sys.path.insert(0, '.')
os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = "1"
from comfyui_propainter_nodes_amd import image_utils, pipeline, synth, weights  # noqa: E402

dev = torch.device('cuda:0')
T = int(os.environ.get("DIAG_T", "60"))
H, W = (int(v) for v in os.environ.get("DIAG_HW", "720,1280").split(","))
NL = int(os.environ.get("DIAG_NL", "20"))
image, mask = synth.synthetic_clip(T, H, W)
fr, fm, md = image_utils.prepare_frames_and_masks(image_utils.image_to_uint8_frames(image), mask, image_utils.ImageConfig(W, H, 5, 8, (W, H), T))
models = pipeline.models_from_state_dicts(weights.synth_state_dicts(0), dev)
models.raft_model.max_volume_bytes = int(os.environ.get("DIAG_MVB_GB", "48")) << 30
cfg = pipeline.ProPainterConfig(10, NL, 80, 20, 'enable', T, dev, (W, H))
ref = None
ALL = ("PP_RAFT_LANES", "PP_ENC_LANES", "PP_FEATPROP_LANES", "PP_WINDOW_LANES")
serial = {k: "1" for k in ALL}
cases = [("all lanes 1", serial), ("all lanes 1 again", serial)]
for k in ALL:
    cases += [(f"only {k}=2", {**serial, k: "2"}), (f"only {k}=2 again", {**serial, k: "2"})]
cases += [("default", {}), ("default again", {})]
if os.environ.get("DIAG_FEATPROP_ONLY"):
    fp = {**serial, "PP_FEATPROP_LANES": "2"}
    cases = [("all lanes 1", serial)] + [(f"featprop lanes, graphs on #{i}", fp) for i in range(3)] + \
            [(f"featprop lanes, PP_GRAPHS=0 #{i}", {**fp, "PP_GRAPHS": "0"}) for i in range(3)] + \
            [(f"featprop lanes, PP_DEFORM_FUSED=force #{i}", {**fp, "PP_DEFORM_FUSED": "force"}) for i in range(2)] + \
            [("serial, PP_DEFORM_FUSED=force", {**serial, "PP_DEFORM_FUSED": "force"})]
for tag, env in cases:
    for k in ALL + ("PP_GRAPHS", "PP_DEFORM_FUSED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    out = pipeline.run_inpainting(models, fr, fm, md, cfg, to_host=False)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    d = (out.int() - ref.int()).abs()
    print(f"{tag:34s}: equal to the serial run: {bool(torch.equal(out, ref))}, differing bytes {int((d > 0).sum())}, max {int(d.max())}", flush=True)
