#!/bin/bash
# r04 final GPU call: the whole -m gpu suite at HEAD, then the profile set (kernel trace, FETCH_SIZE / WRITE_SIZE, bench line),
# then the per-layer table and the cfg 3 / cfg 4 / cfg 5 throughput lines.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 bash tools/profile_bench.sh r04 > $O/profile.log 2>&1; tail -3 $O/profile.log | cut -c1-300
timeout 200 python tools/profile_layers.py > $O/conv_layers.log 2>/dev/null; head -3 $O/conv_layers.log
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
for c in 3 4 5; do timeout 300 python tools/run_config.py --config $c --reps 2 2>&1 | grep -v "done$" | tail -2 | cut -c1-400 >> $O/configs.log; done; cat $O/configs.log | grep "^{" | cut -c1-200
