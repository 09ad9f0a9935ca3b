"""Work-group timeline of conv_gemm_f16_kernel on the transformer Linears (r06).  Needs the trace build of the library:
    bash tools/build_variant.sh gemmtrace conv_gemm_f16 -DPP_GEMM_TRACE     (build container)
    gpurun -- 'python tools/trace_gemm.py'                                    (swaps tools/variants/gemmtrace.so in for the run)
Prints, per layer: launch duration (HIP events), the work-groups' lifetime split (start -> copies issued -> first stage landed ->
loop end -> stores drained), and how many work-groups are alive on XCC 0 over the launch."""
import ctypes
import shutil
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
LIB = ROOT / "comfyui_propainter_nodes_amd" / "libpropainter_mi355.so"
VAR = ROOT / "tools" / "variants" / "gemmtrace.so"


def main():
    shutil.copy(LIB, "/tmp/product.so")
    shutil.copy(VAR, LIB)
    try:
        from comfyui_propainter_nodes_amd import lib, ops
        L = lib.load()
        dev = torch.device("cuda:0")
        for name, cin, cout in (("qkv 512->1536", 512, 1536), ("proj 512->512", 512, 512), ("fc1 512->1960", 512, 1960)):
            x = torch.randn(19, 30, 54, cin, device=dev).half()
            spec = ops.make_conv_spec(torch.randn(cout, cin, 1, 1) * 0.05, torch.zeros(cout), torch.float16).to(dev)
            out = torch.empty(19, 30, 54, cout, device=dev, dtype=torch.float16)
            for _ in range(3):
                ops.conv2d(spec, [x], out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(spec, [x], out)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            M = 19 * 30 * 54
            nwg = -(-M // 128) * -(-cout // 256)
            buf = np.zeros((nwg, 6), dtype=np.uint64)
            rc = L.cdll.pp_debug_gemm_trace(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(nwg))
            assert rc == 0
            t = buf[:, :5].astype(np.int64)
            t0 = t[:, 0].min()
            span = t[:, 4].max() - t0
            tick_ghz = span / us / 1e3
            life = t[:, 4] - t[:, 0]
            ph = [t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]]
            print(f"{name}: {nwg} work-groups, launch {us:.1f} us (events), first start -> last end {span} ticks (= {tick_ghz:.2f} GHz ticks); "
                  f"work-group life mean {life.mean():.0f} / max {life.max()} ticks = {life.mean() / tick_ghz / 1e3:.1f} us; phases mean ticks: "
                  f"setup+issue {ph[0].mean():.0f}, first stage lands {ph[1].mean():.0f}, loop {ph[2].mean():.0f}, epilogue+drain {ph[3].mean():.0f}")
            x0 = buf[:, 5] == 0
            s0, e0_ = np.sort(t[x0, 0] - t0), np.sort(t[x0, 4] - t0)
            marks = np.linspace(0, span, 9)[1:-1]
            alive = [int((s0 <= m).sum() - (e0_ <= m).sum()) for m in marks]
            print(f"   XCC 0: {int(x0.sum())} work-groups; alive at 1/8 .. 7/8 of the launch: {alive}; starts at (us) "
                  f"{[round(float(v) / tick_ghz / 1e3, 1) for v in s0[::max(1, len(s0) // 12)]]}")
    finally:
        shutil.copy("/tmp/product.so", LIB)


if __name__ == "__main__":
    main()
