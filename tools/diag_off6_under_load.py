"""Characterise the load-sensitivity of the f16 halo-tile kernels (r04): the 128 -> 432 fp32-output convolution of the deformable
offsets (and a 128 -> 128 f16 one) under each kernel family, alone and next to a busy stream.  (MI355X; diagnostic)"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402

lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, h, w = 2, 45, 80
x = torch.randn(B, h, w, 128, device=dev, generator=g).half()
big = torch.randn(64, 90, 160, 128, device=dev)
lspec = ops.make_conv_spec(torch.randn(128, 128, 3, 3) * 0.05, torch.zeros(128), torch.float32, padding=1, split=True).to(dev)
lout = torch.empty(64, 90, 160, 128, device=dev)
side = torch.cuda.Stream(dev)
REPS = int(os.environ.get("REPS", "150"))
for cout, odt in ((432, torch.float32), (432, torch.float16), (128, torch.float16), (384, torch.float16)):
    spec = ops.make_conv_spec(torch.randn(cout, 128, 3, 3) * 0.05, torch.randn(cout), torch.float16, padding=1).to(dev)
    for fam, env in (("halo ct", {}), ("halo runtime taps", {"PP_CONV_HALO_CT": "0"}), ("flat", {"PP_CONV_HALO": "0"})):
        for k in ("PP_CONV_HALO_CT", "PP_CONV_HALO"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["PP_CONV_HALO"] = os.environ.get("PP_CONV_HALO", "force")
        os.environ["PP_CONV_KSPLIT"] = "0"
        lib.reload_options()
        ref = torch.empty(B, h, w, cout, device=dev, dtype=odt)
        ops.conv2d(spec, [x], ref)
        torch.cuda.synchronize()
        res = {}
        for mode in ("quiet", "same stream", "two streams"):
            bad = 0
            for rep in range(REPS):
                out = torch.full_like(ref, float("nan"))
                if mode == "two streams":
                    ev = torch.cuda.Event(); ev.record()
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        ops.conv2d(spec, [x], out)
                    ops.conv2d(lspec, [big], lout)
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    if mode == "same stream":
                        ops.conv2d(lspec, [big], lout)
                    ops.conv2d(spec, [x], out)
                torch.cuda.synchronize()
                bad += 0 if torch.equal(out, ref) else 1
            res[mode] = bad
        print(f"3x3 128->{cout} out {str(odt)[6:]:8s} {fam:18s}: differing runs of {REPS}: {res}", flush=True)
