"""Run one BASELINE.json config end to end on the MI355X and print frames/s + stage times + peak memory."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import image_utils, lib, nodes, pipeline, synth, weights  # noqa: E402

CONFIGS = {
    2: dict(T=80, H=360, W=640, nl=10, rs=10, sv=80, iters=20, outpaint=None),
    3: dict(T=80, H=360, W=640, nl=10, rs=10, sv=80, iters=20, outpaint=(1.2, 1.0)),
    # configs[3]: the 640-frame clip of the 8-GPU run, here on ONE GPU: as one process (8 sub-videos one after the other)
    # and as 8 in-process virtual ranks driving the sharded code path (distributed.run_simulated); the two results must
    # be identical (--virtual-ranks N)
    4: dict(T=640, H=360, W=640, nl=10, rs=10, sv=80, iters=20, outpaint=None),
    5: dict(T=160, H=720, W=1280, nl=20, rs=10, sv=80, iters=20, outpaint=None),
}


def parity(case: str, fp16: str, reps: int):
    """Throughput AND parity of one run: the fixture's clip through OUR node method, compared with the output of the reference's
    node method on that clip (CPU fp32, minted by tests/golden/make_golden.py)."""
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
    from node_case import evaluate_node_case

    lib.load()
    os.environ["PP_ALLOW_SYNTHETIC_WEIGHTS"] = "1"
    best = []
    for rep in range(reps):
        secs = []
        m = evaluate_node_case(case, fp16, check=False, timer=secs.append)
        best.append(secs[0])
    m["node_call_seconds_traced"] = round(min(best), 3)
    m["node_call_frames_per_s_traced"] = round(m["frames"] / min(best), 2)
    m["note"] = ("the node call of this leg keeps its stage tensors for the comparison (nodes.TRACE): the un-traced frames/s of the same "
                 "configuration is the line printed without --parity")
    print(json.dumps(m), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--size", type=str, default="")
    ap.add_argument("--virtual-ranks", type=int, default=0, help="also run the sharded driver with N in-process ranks and compare")
    ap.add_argument("--parity", type=str, default="", help="a reference-minted node fixture (tests/golden/<name>.npz): run ITS clip through "
                    "the node method and print PSNR / max LSB / mask + flow agreement beside the frames/s of that call")
    ap.add_argument("--fp16", type=str, default="enable")
    a = ap.parse_args()
    if a.parity:
        return parity(a.parity, a.fp16, a.reps)
    c = dict(CONFIGS[a.config])
    if a.frames:
        c["T"] = a.frames
    if a.size:
        c["W"], c["H"] = [int(v) for v in a.size.split("x")]
    lib.load()
    dev = torch.device("cuda:0")
    os.environ["PP_TIMING"] = "1"
    sds, prov = weights.get_state_dicts(0)
    models = pipeline.models_from_state_dicts(sds, dev)
    image, mask = synth.synthetic_clip(c["T"], c["H"], c["W"])
    u8 = image_utils.image_to_uint8_frames(image)
    if c["outpaint"]:
        oc = image_utils.ImageOutpaintConfig(c["W"], c["H"], 5, 8, (c["W"], c["H"]), c["T"], *c["outpaint"])
        fr, fm, md = image_utils.extrapolation(u8, oc)
        size = oc.outpaint_size
    else:
        ic = image_utils.ImageConfig(c["W"], c["H"], 5, 8, (c["W"], c["H"]), c["T"])
        fr, fm, md = image_utils.prepare_frames_and_masks(u8, mask, ic)
        size = ic.process_size
    cfg = pipeline.ProPainterConfig(c["rs"], c["nl"], c["sv"], c["iters"], "enable", c["T"], dev, size)
    args = (models, torch.from_numpy(fr).to(dev), torch.from_numpy(fm).to(dev), torch.from_numpy(md).to(dev), cfg)
    for rep in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipeline.run_inpainting(*args, to_host=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": a.config, "frames": c["T"], "size": list(size), "seconds": round(dt, 3),
                          "frames_per_s": round(c["T"] / dt, 2), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                          "out_mean": float(out.float().mean())}), flush=True)
    if a.virtual_ranks:
        from comfyui_propainter_nodes_amd import distributed as D

        os.environ["PP_TIMING"] = "0"
        os.environ["PP_GRAPHS"] = "0"   # virtual ranks share one model object (and its captured graphs' static buffers)
        single = pipeline.run_inpainting(*args, to_host=False)
        torch.cuda.reset_peak_memory_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = D.run_simulated(lambda r: D.GpuBackend(models, cfg), a.virtual_ranks, cfg, *args[1:4])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = [bool(torch.equal(r, single)) for r in res]
        diff = max(int((r.int() - single.int()).abs().max()) for r in res)
        print(json.dumps({"config": a.config, "virtual_ranks": a.virtual_ranks, "frames": c["T"], "seconds_all_ranks_serialised": round(dt, 3),
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                          "every_rank_bit_identical_to_single_process": all(same), "max_abs_diff_u8": diff}), flush=True)


if __name__ == "__main__":
    main()
