#!/bin/bash
# First GPU call of the next round (gpurun): the whole -m gpu suite at HEAD, then the A/B of the flat-tile order that was
# committed after round 3's last GPU minute, then the profile set (kernel trace, FETCH_SIZE / WRITE_SIZE, bench line).
#   gpurun --timeout 2400 -- 'bash tools/gpu_r4_first.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_first; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 bash tools/ab_conv_order.sh > $O/ab_conv_order.log 2>&1; tail -40 $O/ab_conv_order.log | cut -c1-150
timeout 60 python tools/bench_deform.py > $O/bench_deform.jsonl 2>&1
timeout 900 bash tools/profile_bench.sh r04 > $O/profile.log 2>&1; tail -20 $O/profile.log | cut -c1-200
