#!/bin/bash
# Per-step cycle trace of one work-group of window_attention_f16_kernel (s_memtime stamps of waves 0 and 3 at the phase
# boundaries of every 32-key step), from a -DPP_ATTN_TRACE build of that one translation unit linked with the product objects.
#   tools/trace_attention.sh --build     (CPU box)      gpurun -- 'bash tools/trace_attention.sh'   (MI355X)
cd "$(dirname "$0")/.."
PKG=comfyui_propainter_nodes_amd
D=tools/ablate/attn_trace
if [ "${1:-}" = "--build" ]; then
  mkdir -p $D
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPP_ATTN_TRACE -I $PKG/csrc -I include -c $PKG/csrc/window_attention.hip -o $D/window_attention.o || exit 1
  objs=$(ls $PKG/build/hip/*.o | grep -v "window_attention.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $D/window_attention.o -o $D/libpropainter_mi355.so && rm $D/window_attention.o
  ls -la $D; exit 0
fi
for m in 16 8; do PP_LIB=$D/libpropainter_mi355.so python tools/bench_attention.py --masked $m --reps 4 2>&1 | grep -v amdgpu.ids; done
