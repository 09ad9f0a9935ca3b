"""Enumerate every pp_conv2d launch of one clip WITHOUT running a kernel: the host pipeline is driven with a recording
stand-in for the library (tensors are allocated but never touched), so the shape mix of BASELINE.json cfg 2 can be
tabulated on a machine without a GPU.  Used to decide which tile configurations matter.

    python tools/dry_run_shapes.py [--frames 80] [--size 640x360]"""
import argparse
import ctypes
import sys
from collections import defaultdict
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from comfyui_propainter_nodes_amd import lib, pipeline, weights  # noqa: E402


class Recorder:
    is_emulator = True  # accept CPU tensors

    def __init__(self):
        self.convs = defaultdict(lambda: [0, 0.0])
        self.others = defaultdict(int)

    def call(self, fname, stream, params):
        if fname != "pp_conv2d":
            self.others[fname] += 1
            return
        p = params
        cin = sum(int(p.in_C[s]) for s in range(p.nseg))
        m = int(p.N * p.Ho * p.Wo)
        key = (int(p.dtype), int(p.Cout), cin, int(p.kh), int(p.kw), int(p.Z), m, int(p.sh))
        self.convs[key][0] += 1
        self.convs[key][1] += 2.0 * m * p.Cout * cin * p.kh * p.kw * p.Z


def tile_of(dtype, cout, m, z):
    blocks128 = ((m + 127) // 128) * ((cout + 127) // 128) * z
    small = blocks128 < 224
    w256 = (cout + 255) // 256 * 256 - cout
    xl = ((cout + 255) // 256 * 256 == (cout + 127) // 128 * 128) and w256 * 8 <= cout and blocks128 >= 1024
    if cout > 64:
        if small:
            t = "128x32"
        else:
            w128, w96 = (cout + 127) // 128 * 128 - cout, (cout + 95) // 96 * 96 - cout
            t = "96x128" if w96 + 32 <= w128 else "128x128"
    elif cout > 32:
        t = "64x32" if small else "64x128"
    elif cout > 16:
        t = "32x128"
    else:
        t = "16x256"
    return t, xl, blocks128


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--size", default="640x360")
    a = ap.parse_args()
    W, H = [int(v) for v in a.size.split("x")]
    rec = Recorder()
    lib._lib = rec
    dev = torch.device("cpu")
    sds, _ = weights.get_state_dicts(0)
    models = pipeline.models_from_state_dicts(sds, dev)
    fr, fm, md = bench.make_inputs(a.frames, H, W, 5, 8)
    cfg = pipeline.ProPainterConfig(10, 10, 80, 20, "enable", a.frames, dev, (W, H))
    pipeline.run_inpainting(models, torch.from_numpy(fr), torch.from_numpy(fm), torch.from_numpy(md), cfg, to_host=False)
    names = {0: "f32", 1: "f16", 4: "f32x2"}
    rows = sorted(rec.convs.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for v in rec.convs.values())
    per_dtype = defaultdict(float)
    xl_flops = defaultdict(float)
    print("| dtype | Cout | Cin | k | Z | M | launches | TFLOP | share | tile | 8-wave eligible |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for (dt, cout, cin, kh, kw, z, m, st), (n, fl) in rows:
        t, xl, _ = tile_of(dt, cout, m, z)
        per_dtype[names[dt]] += fl
        if xl:
            xl_flops[names[dt]] += fl
        if fl / total >= 0.004:
            print(f"| {names[dt]} | {cout} | {cin} | {kh}x{kw}{'/s' + str(st) if st > 1 else ''} | {z} | {m} | {n} | {fl / 1e12:.2f} | "
                  f"{100 * fl / total:.1f} % | {t} | {'yes' if xl else ''} |")
    print()
    for k, v in per_dtype.items():
        print(f"{k}: {v / 1e12:.1f} TFLOP per clip, {100 * xl_flops[k] / v:.0f} % of it on shapes eligible for the 8-wave tiles")
    print("other launches:", dict(rec.others))


if __name__ == "__main__":
    main()
