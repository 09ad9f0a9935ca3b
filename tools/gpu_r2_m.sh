#!/bin/bash
# fast epilogue: conv micro-benchmarks (all shapes), tall vs 8-row, conv parity on the hardware, bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2m; mkdir -p $O
echo "== default (tall off unless eligible)"; PP_CONV_HALO_TALL=0 timeout 120 tools/convbench 2>&1 | tee $O/convbench.log
S="raft_gru_1x5_f32x2 raft_gru128_1x5_f32x2 raft_convc2_f32x2 raft_fh1_f32x2"
echo "== tall"; PP_CONV_HALO_TALL=force timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_tall.log
timeout 900 python -m pytest tests/test_conv.py -m gpu -q 2>&1 | tail -5
PP_CONV_HALO_TALL=0 PP_TIMING=1 timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; grep -E "stage ms" $O/bench.log | tail -1; tail -1 $O/bench.log | cut -c1-300
