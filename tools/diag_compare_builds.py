"""r06 diagnostic: do two builds of the library give the same bits on the same clip (serial schedule)?  Usage:
   python tools/diag_compare_builds.py dump <tag> [path/to/lib.so]    -> /tmp/build_cmp_<tag>.pt (stage tensors of one traced pass)
   python tools/diag_compare_builds.py diff <tagA> <tagB>"""
import os
import shutil
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = "1"
for k in ("PP_RAFT_LANES", "PP_ENC_LANES", "PP_FEATPROP_LANES", "PP_WINDOW_LANES"):
    os.environ[k] = "1"
LIB = ROOT / "comfyui_propainter_nodes_amd" / "libpropainter_mi355.so"

if sys.argv[1] == "dump":
    tag = sys.argv[2]
    swap = len(sys.argv) > 3
    if swap:
        shutil.copy(LIB, "/tmp/product_keep.so")
        shutil.copy(sys.argv[3], LIB)
    try:
        from comfyui_propainter_nodes_amd import image_utils, pipeline, synth, weights
        dev = torch.device("cuda:0")
        T, H, W = 40, 360, 640
        image, mask = synth.synthetic_clip(T, H, W)
        fr, fm, md = image_utils.prepare_frames_and_masks(image_utils.image_to_uint8_frames(image), mask, image_utils.ImageConfig(W, H, 5, 8, (W, H), T))
        models = pipeline.models_from_state_dicts(weights.synth_state_dicts(0), dev)
        cfg = pipeline.ProPainterConfig(10, 10, 80, 20, "enable", T, dev, (W, H))
        tr, wtr, seen = {}, {}, {}
        gen = models.inpaint_model
        fw0, pw0, pc0 = gen.forward_window, gen.propagate_windows, gen.prepare_clip

        def prepare_clip(*a, **k):          # stage tensors of the generator: encoder, feature propagation, window 0's transformer
            st = pc0(*a, **k)
            seen["enc"] = st.enc.clone()
            return st

        def propagate_windows(st, windows):
            props = pw0(st, windows)
            seen["prop0"] = props[0].clone()
            return props

        def forward_window(st, nb, refs, trace=None, **k):
            if "done" not in seen:
                seen["done"] = True
                return fw0(st, nb, refs, trace=wtr, **k)
            return fw0(st, nb, refs, trace=trace, **k)

        gen.forward_window, gen.propagate_windows, gen.prepare_clip = forward_window, propagate_windows, prepare_clip
        comp = pipeline.run_inpainting(models, fr, fm, md, cfg, trace=tr)
        torch.save({"gt": tr["gt_flows"].cpu(), "pred": tr["pred_flows"].cpu(), "upd": tr["updated_frames"].cpu(),
                    "enc": seen["enc"].cpu(), "prop0": seen["prop0"].cpu(), "tok0": wtr["tok"].cpu(), "tok_out0": wtr["tok_out"].cpu(),
                    "enc3_0": wtr["enc3"].cpu(), "img0": tr["pred_imgs"][0],
                    "img3": tr["pred_imgs"][3], "comp": comp.cpu()}, f"/tmp/build_cmp_{tag}.pt")
        print("dumped", tag)
    finally:
        if swap:
            shutil.copy("/tmp/product_keep.so", LIB)
else:
    a, b = torch.load(f"/tmp/build_cmp_{sys.argv[2]}.pt"), torch.load(f"/tmp/build_cmp_{sys.argv[3]}.pt")
    for k in a:
        d = (a[k].float() - b[k].float()).abs()
        print(f"{k:8s}: equal {bool(torch.equal(a[k], b[k]))}, differing {int((d > 0).sum())} of {d.numel()}, max {float(d.max()):.3e}")
