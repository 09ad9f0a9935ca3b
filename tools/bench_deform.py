"""A/B of the modulated deformable convolution on the MI355X (run via gpurun): the two-launch form (pp_deform_cols + 1x1
pp_conv2d over the 9*Cin columns) against the one-launch pp_deform_conv, pixel blocks in launch order and in XCD-contiguous
order (PP_DEFORM_XCD), at the two shapes of the pipeline for BASELINE configs[1].  With tools/experiments/deform_conv_forms.patch
applied the library also has the forms that were measured and not shipped (PP_DEFORM_TILE=16 / 32 / ksplit / ksplit1; without
the patch the knob is ignored and every row times the shipped form) -- profiles/r03_deform_fusion.md:

  featprop  feature propagation of the inpainting generator: nw windows x 90 x 160 pixels, 128 channels, flow added to the offsets
  rfc       flow completion: 2 x 45 x 80 pixels, two 128-channel inputs

Prints one JSON line per (shape, form): us per call (HIP events over `reps` back-to-back calls), whether the fused result
equals the two-launch result bit for bit, and the launch-level algorithmic GFLOP."""
import argparse
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from comfyui_propainter_nodes_amd import lib, ops  # noqa: E402


def timed(fn, reps):
    if not torch.cuda.is_available():   # --emu dry run: wall clock of the emulator, meaningless as a timing
        import time
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) * 1e6 / reps
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--emu", action="store_true", help="dry run of this script on CPU under the x86 emulator (tiny shapes)")
    args = ap.parse_args()
    g = torch.Generator().manual_seed(7)
    shapes = {"featprop": (args.windows, 90, 160, 128, 0, True), "rfc": (2, 45, 80, 128, 128, False)}
    if args.emu:
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests" / "emu"))
        import emu_loader

        emu_loader.load_emulator()
        dev = torch.device("cpu")
        shapes = {"featprop": (1, 10, 12, 128, 0, True), "rfc": (1, 9, 8, 128, 128, False)}
    else:
        lib.load()
        dev = torch.device("cuda:0")
    for name, (n, h, w, c0, c1, with_flow) in shapes.items():
        cin, dg, cout = c0 + c1, 16, 128
        x = (torch.randn(n, h, w, cin, generator=g) * 0.5).half().to(dev)
        x0, x1 = x[..., :c0], (x[..., c0:] if c1 else None)
        om = torch.cat([torch.randn(n, h, w, 2 * dg * 9, generator=g) * 1.5, torch.rand(n, h, w, dg * 9, generator=g)], -1).to(dev)
        flow = (torch.randn(n, h, w, 2, generator=g) * 2).to(dev) if with_flow else None
        spec = ops.make_conv_spec(torch.randn(cout, 9 * cin, 1, 1, generator=g) * 0.03, torch.randn(cout, generator=g),
                                  torch.float16).to(dev)
        cols = torch.empty(n, h, w, 9 * cin, device=dev, dtype=torch.float16)
        two = torch.empty(n, h, w, cout, device=dev, dtype=torch.float16)
        one = torch.empty_like(two)
        gflop = 2.0 * n * h * w * cout * 9 * cin / 1e9

        def two_launch():
            ops.deform_cols(x0, x1, om, cols, dg=dg, flow=flow)
            ops.conv2d(spec, [cols], two)

        def knobs(**kw):
            for k_, v in kw.items():
                if v is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v
            lib.reload_options()

        cols_ref = None
        for xcd in ("0", "1"):
            knobs(PP_DEFORM_XCD=xcd)
            us_cols = timed(lambda: ops.deform_cols(x0, x1, om, cols, dg=dg, flow=flow), args.reps)
            us_two = timed(two_launch, args.reps)
            same = True if cols_ref is None else bool(torch.equal(cols, cols_ref))
            cols_ref = cols.clone() if cols_ref is None else cols_ref
            print(json.dumps({"shape": name, "form": "two launches (pp_deform_cols + pp_conv2d)", "xcd_order": xcd, "us": round(us_two, 1),
                              "us_deform_cols_alone": round(us_cols, 1), "same_columns_as_launch_order": same,
                              "gflop": round(gflop, 2), "pixels": n * h * w, "cin": cin}), flush=True)
            if xcd == "0":
                us_base = us_two
            for tile in ("16", "32", "ksplit", "ksplit1"):
                knobs(PP_DEFORM_TILE=tile)
                us = timed(lambda: ops.deform_conv(spec, x0, x1, om, one, dg=dg, flow=flow), args.reps)
                print(json.dumps({"shape": name, "form": f"pp_deform_conv tile={tile}", "xcd_order": xcd, "us": round(us, 1),
                                  "speedup_vs_two_launches": round(us_base / us, 2),
                                  "bit_identical_to_two_launches": bool(torch.equal(one, two)),
                                  "max_abs_diff": float((one.float() - two.float()).abs().max())}), flush=True)
            knobs(PP_DEFORM_TILE=None)
        knobs(PP_DEFORM_XCD=None, PP_DEFORM_TILE=None)


if __name__ == "__main__":
    main()
