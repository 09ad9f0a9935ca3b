#!/bin/bash
# s_memtime phase trace of the PP_F32X2 halo kernels (build: bash tools/build_variant.sh trace conv_halo -DPP_HALO_TRACE):
# per tap copy issue / fragment reads + MFMAs / counted waits + barrier, chunk boundary, prologue, epilogue with its own stamps.
#   gpurun -- 'bash tools/trace_halo.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R5_OUT:-halo_trace}; mkdir -p $O
L=comfyui_propainter_nodes_amd/libpropainter_mi355.so
cp $L /tmp/product.so; cp tools/variants/trace.so $L
for solo in 0 1; do PP_HALO_TRACE_SOLO=$solo timeout 60 tools/convbench ${SHAPES:-raft_gru_1x5_f32x2 raft_gru128_5x1_f32x2 raft_convc2_f32x2 raft_fh1_f32x2} 2>&1 | grep -v "^$"; done | tee $O/halo_trace.log
cp /tmp/product.so $L
