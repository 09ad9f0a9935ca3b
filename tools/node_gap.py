"""Where does the node call lose time against the device-resident step?  (MI355X; diagnostic for SURVEY.md 8 f3)
Times, best of 3 each: the bench-style device-resident step, then the node method with the three output modes."""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = "1"
import bench  # noqa: E402
from comfyui_propainter_nodes_amd import lib, nodes, pipeline, synth, weights  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
C = bench.CFG
sds, _ = weights.get_state_dicts(0)
models = pipeline.initialize_models(dev, "enable")
frames_u8, fm, md = bench.make_inputs(C["T"], C["H"], C["W"], C["mask_dilates"], C["flow_mask_dilates"])
cfg = pipeline.ProPainterConfig(C["ref_stride"], C["neighbor_length"], C["subvideo_length"], C["raft_iter"], "enable", C["T"], dev,
                                (C["W"], C["H"]))
fr_d, fm_d, md_d = (torch.from_numpy(a).to(dev) for a in (frames_u8, fm, md))


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        out = None                      # (freeing the previous 221 MB IMAGE is the caller's time)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


print(f"device-resident step: {best(lambda: pipeline.run_inpainting(models, fr_d, fm_d, md_d, cfg, to_host=False)):.1f} ms")
image, mask = synth.synthetic_clip(C["T"], C["H"], C["W"], 1234)
node = nodes.ProPainterInpaint()
args = (image, mask, C["W"], C["H"], C["mask_dilates"], C["flow_mask_dilates"], C["ref_stride"], C["neighbor_length"],
        C["subvideo_length"], C["raft_iter"], "enable")
for mode in os.environ.get("NODE_GAP_MODES", "device,host").split(","):
    os.environ["PP_OUTPUT"] = mode.split("-")[0]
    nodes._HostImageSink.wait_mode = "sync" if mode.endswith("sync") else "poll"
    nodes._Timer.collect = False
    t = best(lambda: node.propainter_inpainting(*args))
    nodes._Timer.collect = True
    node.propainter_inpainting(*args)
    synced = dict(nodes._Timer.last)
    nodes._Timer.sync = False
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = node.propainter_inpainting(*args)
    torch.cuda.synchronize()
    t1 = (time.perf_counter() - t0) * 1e3
    del keep
    nodes._Timer.sync = True
    print(f"node call, PP_OUTPUT={mode}: {t:.1f} ms; stages (with stage syncs): {synced}; host-only stamps of a {t1:.1f} ms call: {nodes._Timer.last}")
