"""r06 experiment: the transformer GEMMs with some work-groups of the first round delayed (tools/variants/dephase.so, built by
`tools/build_variant.sh dephase conv_gemm_f16 -DPP_GEMM_DEPHASE`): does de-phasing the two work-groups of a CU shorten the launch?"""
import ctypes
import shutil
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
LIB = ROOT / "comfyui_propainter_nodes_amd" / "libpropainter_mi355.so"
shutil.copy(LIB, "/tmp/product_keep.so")
shutil.copy(ROOT / "tools" / "variants" / "dephase.so", LIB)
try:
    from comfyui_propainter_nodes_amd import lib, ops
    L = lib.load()
    cd = L.cdll
    cd.pp_debug_gemm_dephase.argtypes = [ctypes.c_int32] * 4
    dev = torch.device("cuda:0")
    for name, M, K, Cout in (("qkv", 30780, 512, 1536), ("proj", 30780, 512, 512), ("fc1", 30780, 512, 1960), ("fc2 (materialised)", 30780, 1960, 512)):
        x = torch.randn(1, 1, M, K, device=dev).half()
        w = torch.randn(Cout, K, 1, 1) * 0.05
        spec = ops.make_conv_spec(w, torch.zeros(Cout), torch.float16).to(dev)
        out = torch.empty(1, 1, M, Cout, device=dev, dtype=torch.float16)
        ref = None
        for lo, hi, ticks, mod in ((0, 0, 0, 0), (256, 512, 6000, 0), (256, 512, 12000, 0), (256, 512, 20000, 0), (256, 512, 30000, 0),
                                   (0, 512, 12000, 1), (0, 512, 12000, 8), (0, 100000, 12000, 256), (0, 512, 12000, 2), (0, 512, 20000, 16)):
            assert cd.pp_debug_gemm_dephase(lo, hi, ticks, mod) == 0
            for _ in range(3):
                ops.conv2d(spec, [x], out)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            assert torch.equal(ref, out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                ops.conv2d(spec, [x], out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            print(f"{name:20s} late = [{lo}, {hi}) mod {mod:3d}, {ticks:6d} ticks: {ms * 1e3:6.1f} us  {2.0 * M * K * Cout / ms / 1e9:5.0f} TF/s", flush=True)
finally:
    shutil.copy("/tmp/product_keep.so", LIB)
