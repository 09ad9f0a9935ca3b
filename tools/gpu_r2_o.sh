#!/bin/bash
# 4-op operand split: conv micro-benchmarks, conv / RAFT / e2e parity on the hardware, bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2o; mkdir -p $O
timeout 120 tools/convbench raft_gru_1x5_f32x2 raft_gru128_1x5_f32x2 raft_gru128_5x1_f32x2 raft_convc2_f32x2 raft_fh1_f32x2 2>&1 | tee $O/convbench.log
timeout 1200 python -m pytest tests/test_conv.py tests/test_raft.py tests/test_e2e.py tests/test_baseline_configs.py -m gpu -q -x 2>&1 | tail -5
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
