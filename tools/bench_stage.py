"""Time individual stages at BASELINE cfg 2 on the MI355X (run via gpurun)."""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, raft, synth, weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="raft")
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    lib.load()
    sds = weights.synth_state_dicts(0)
    image, mask = synth.synthetic_clip(args.frames, 360, 640)
    frames = (image * 2 - 1).cuda()
    if args.stage == "raft":
        R = raft.RaftFlow(sds["raft"], "cuda:0")
        for rep in range(2):
            torch.cuda.synchronize()
            t = time.time()
            ff, fb = R(frames, args.iters)
            torch.cuda.synchronize()
            print(f"raft {args.frames} frames, {args.iters} iters: {time.time() - t:.3f} s  flow absmax {ff.abs().max().item():.2f}",
                  flush=True)
        print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)


if __name__ == "__main__":
    main()
