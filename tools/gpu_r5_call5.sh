#!/bin/bash
# (r05 end: the build hooks this script times -- PP_HALO_TRIM64 / PP_HALO_WMID / PP_EPI_STORE_MODE -- were measured and removed from csrc/;
#  TRIM64 is archived in tools/experiments/r05_halo_hooks.patch, the other two are in git history.  Kept as the record of how
#  profiles/r05_ab_*.log and r05_halo_trace_call1_r04_kernel.log were produced.)
# r05 GPU call 5: the store path in isolation (tools/probes/store_rate) and the cache policy of the epilogue's stores (variants st1-st4 =
# -DPP_EPI_STORE_MODE=1..4 + trace) on the PP_F32X2 halo kernels
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call5; mkdir -p $O
L=comfyui_propainter_nodes_amd/libpropainter_mi355.so
timeout 120 tools/probes/store_rate 2>&1 | tee $O/store_rate.log
cp $L /tmp/product.so
for v in trace st1 st2 st3 st4; do cp tools/variants/$v.so $L; echo "== $v"; timeout 60 tools/convbench raft_gru_1x5_f32x2 raft_convc2_f32x2 2>&1; done | tee $O/store_policy.log
cp /tmp/product.so $L
