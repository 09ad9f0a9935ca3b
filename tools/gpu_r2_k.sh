#!/bin/bash
# 16-row one-wave-per-SIMD halo kernel: A/B against the 8-row kernels, hardware parity
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2k; mkdir -p $O
S="raft_gru_1x5_f32x2 raft_gru128_1x5_f32x2 raft_gru128_5x1_f32x2 raft_convc2_f32x2 raft_fh1_f32x2"
echo "== tall"; PP_CONV_HALO_TALL=force timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_tall.log
echo "== 8-row (PP_CONV_HALO_TALL=0)"; PP_CONV_HALO_TALL=0 timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_8row.log
timeout 600 python -m pytest tests/test_conv.py -m gpu -q -k "tall" 2>&1 | tail -30
