"""Ablation timing of the convolution kernels (profiling aid, not part of the product).

    python tools/ablate_conv.py --build     # here (CPU): variant libraries tools/ablate/libpp_abl_<mask>.so
    python tools/ablate_conv.py             # on the MI355X (gpurun): time every variant, write gpurun_out/ablate.json

Each variant is the product library with conv_igemm.hip / conv_split.hip compiled with -DPP_ABLATE=<mask> (1 no MFMA, 2 no pixel
loads, 4 no weight loads, 8 no operand-split arithmetic, 16 no LDS fragment reads, 32 no K-loop barriers) from a scratch copy of
csrc/ that tools/ablate_src.sh patches with tools/ablate_hooks.patch: the product sources carry no ablation code.
The results of an ablated kernel are meaningless; only the time differences are read."""
import json
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tools" / "ablate"
MASKS = [1, 2, 4, 6, 8, 16, 32, 63]
sys.path.insert(0, str(ROOT))


def build():
    from comfyui_propainter_nodes_amd import build as B

    B.build_hip()
    src = Path(subprocess.run([str(ROOT / "tools" / "ablate_src.sh")], check=True, capture_output=True, text=True).stdout.strip())
    conv = ("conv_igemm", "conv_split")
    objs = [o for o in (B.PKG / "build" / "hip").glob("*.o") if o.stem not in conv]

    def one(mask):
        mine = []
        for name in conv:
            obj = OUT / f"{name}_{mask}.o"
            subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-DPP_ABLATE={mask}",
                            "-I", str(src), "-I", str(ROOT / "include"), "-c", str(src / f"{name}.hip"), "-o", str(obj)],
                           check=True)
            mine.append(obj)
        so = OUT / f"libpp_abl_{mask}.so"
        subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), *map(str, mine), "-o", str(so)],
                       check=True)
        for obj in mine:
            obj.unlink()
        return so

    OUT.mkdir(exist_ok=True)
    with ThreadPoolExecutor(4) as ex:
        for so in ex.map(one, MASKS):
            print("built", so)


SHAPES = [
    # name, split, f16, N, H, W, segC, Cout, k, pad
    ("raft_gru_1x5 split", True, False, 158, 45, 80, [128, 128], 256, (1, 5), (0, 2)),
    ("enc_3x3_256_384 f16", False, True, 16, 90, 160, [256], 384, (3, 3), (1, 1)),
    ("fc1_1x1 f16", False, True, 1, 1, 29160, [512], 1960, (1, 1), (0, 0)),
]


def run_one(mask: int) -> None:
    """Time every shape with one variant library (child process: a faulting variant must not take the others down)."""
    import torch

    from comfyui_propainter_nodes_amd import lib, ops

    dev = torch.device("cuda:0")
    path = lib.HIP_LIB if mask == 0 else OUT / f"libpp_abl_{mask}.so"
    lib._lib = lib.Library(path, is_emulator=False)
    row = {}
    for name, split, f16, N, H, W, segC, Cout, k, p in SHAPES:
        dt = torch.float16 if f16 else torch.float32
        x = [torch.randn(N, H, W, c, device=dev).to(dt) for c in segC]
        w = torch.randn(Cout, sum(segC), *k) * 0.05
        spec = ops.make_conv_spec(w, torch.zeros(Cout), dt, padding=p, seg_channels=segC, split=split).to(dev)
        out = torch.empty(N, *spec.out_hw(H, W), Cout, device=dev, dtype=dt)
        for _ in range(3):
            ops.conv2d(spec, x, out, act="relu")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(spec, x, out, act="relu")
        e1.record()
        torch.cuda.synchronize()
        row[name] = round(e0.elapsed_time(e1) / 10, 4)
    print("ABLATE_ROW " + json.dumps({"mask": mask, "ms": row}), flush=True)


def run():
    res = {}
    masks = [0] + [m for m in MASKS if (OUT / f"libpp_abl_{m}.so").exists()]
    for mask in masks:
        r = subprocess.run([sys.executable, __file__, "--one", str(mask)], capture_output=True, text=True, timeout=120)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("ABLATE_ROW ")]
        res[str(mask)] = json.loads(rows[-1][len("ABLATE_ROW "):])["ms"] if rows else {"error": (r.stderr or r.stdout)[-300:]}
        print(mask, res[str(mask)], flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "ablate.json").write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    elif "--one" in sys.argv:
        run_one(int(sys.argv[sys.argv.index("--one") + 1]))
    else:
        run()
