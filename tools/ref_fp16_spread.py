"""What does the REFERENCE's own fp16 mode do to the generator output?  BUILD-CONTAINER ONLY (imports /root/reference).

The node fixtures are the reference's fp32 CPU run; our fp16 "enable" mode stores the generator's activations as f16 like the
reference's `.half()` generator (utils/model_utils.py:57-58).  tools/diag_lsb_outliers.py --explain shows that the handful of
bytes beyond 2 LSB (cfg3_80f: 5 of 11 M, cfg5_160f: 1 of 15 M) are float differences of 3-4 LSB in ONE window's output, not
truncation cascades.  This script measures the same quantity for the reference against itself: it runs the reference's node on
the fixture's clip and, for the chosen windows, evaluates the reference's InpaintGenerator twice on identical fp32 inputs --
as is (fp32) and as a `.half()` copy (PyTorch CPU half kernels: f16 storage, the reference's fp16 arithmetic) -- and reports
max |half - fp32| in pixel units at the masked pixels of the local frames, the number of values >= 1e-2, and the values at the
positions of OUR outliers (gpurun_out/outliers_<case>_enable.npz).  Other windows are skipped (their output is not needed).

    python tools/ref_fp16_spread.py --case cfg3_80f_node --windows 5,8
"""
from __future__ import annotations

import argparse
import copy
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))

import make_golden as MG  # noqa: E402
from comfyui_propainter_nodes_amd import pipeline, synth, weights  # noqa: E402


class _Stop(Exception):
    pass


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--windows", required=True)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    wins = sorted(int(v) for v in a.windows.split(","))
    kw = dict(MG.NODE_CASES[a.case])
    kind, T, H, W = kw.pop("kind"), kw.pop("T"), kw.pop("H"), kw.pop("W")
    sds = weights.synth_state_dicts(kw.pop("seed", 0), kw.pop("weights_variant", ""))
    ref, models = MG.build_reference_models(sds)
    import reference.propainter_inference as PI
    import reference.propainter_nodes as RN

    gen = models.inpaint_model
    half = copy.deepcopy(gen).half()
    orig_fwd, orig_pi, orig_init = gen.forward, RN.process_inpainting, RN.initialize_models
    cap: dict = {"wi": 0}
    outl = ROOT / "gpurun_out" / f"outliers_{a.case}_enable.npz"
    outliers = np.load(outl) if outl.exists() else None
    t0 = time.time()

    def pi(models_, frames, flow_masks, masks_dilated, config):
        cap["md"] = masks_dilated[0, :, 0].numpy().astype(bool)
        cap["schedule"] = pipeline.window_schedule(pipeline.ProPainterConfig(
            config.ref_stride, config.neighbor_length, config.subvideo_length, config.raft_iter, "disable",
            config.video_length, torch.device("cpu"), config.process_size))
        return PI.process_inpainting(models_, frames, flow_masks, masks_dilated, config)

    def fwd(frames, flows, masks, upd, lt):
        wi = cap["wi"]
        cap["wi"] += 1
        h, w = cap["md"].shape[1:]
        if wi not in wins:
            return torch.zeros(1, lt, 3, h, w)
        nb = cap["schedule"][wi][0]
        out32 = orig_fwd(frames, flows, masks, upd, lt)
        out16 = half(frames.half(), (flows[0].half(), flows[1].half()), masks.half(), upd.half(), lt).float()
        a32 = out32.reshape(-1, 3, h, w).permute(0, 2, 3, 1)
        a16 = out16.reshape(-1, 3, h, w).permute(0, 2, 3, 1)
        worst, nbad, ntot = 0.0, 0, 0
        for i, g in enumerate(nb):
            sel = torch.from_numpy(cap["md"][g])
            d = (a32[i][sel] - a16[i][sel]).abs() * 0.5
            worst, nbad, ntot = max(worst, float(d.max())), nbad + int((d >= 1e-2).sum()), ntot + d.numel()
        print(f"window {wi} ({time.time() - t0:.0f} s): reference half vs reference fp32 at {ntot} masked values: max abs "
              f"{worst:.3e} pixel units = {worst * 255:.2f} LSB, {nbad} values >= 1e-2", flush=True)
        if outliers is not None:
            for (t, y, x, c), vis in zip(outliers["pos"], outliers["visits"]):
                for vw, vi in eval(str(vis)):
                    if vw == wi:
                        print(f"   our outlier at frame {t} ({y},{x}) ch {c}: reference fp32 {float((a32[vi][y, x, c] + 1) / 2 * 255):.3f}, "
                              f"reference half {float((a16[vi][y, x, c] + 1) / 2 * 255):.3f} (x255)")
        if wi >= wins[-1]:
            raise _Stop()
        return out32

    RN.process_inpainting = pi
    RN.initialize_models = lambda device, fp16: models
    gen.forward = fwd
    image, mask = synth.synthetic_clip(T, H, W)
    common = dict(mask_dilates=kw.get("mask_dilates", 5), flow_mask_dilates=kw.get("flow_mask_dilates", 8), ref_stride=kw["ref_stride"],
                  neighbor_length=kw["neighbor_length"], subvideo_length=kw["subvideo_length"], raft_iter=kw["raft_iter"], fp16="disable")
    try:
        if kind == "inpaint":
            RN.ProPainterInpaint().propainter_inpainting(image, mask, kw["width"], kw["height"], **common)
        else:
            RN.ProPainterOutpaint().propainter_outpainting(image, kw["width"], kw["height"], kw.get("width_scale", 1.2),
                                                          kw.get("height_scale", 1.0), **common)
    except _Stop:
        pass
    finally:
        RN.process_inpainting, RN.initialize_models, gen.forward = orig_pi, orig_init, orig_fwd


if __name__ == "__main__":
    main()
