#!/bin/bash
# attention with prefetched K/V tiles: parity, kernel time, bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2s; mkdir -p $O; S=/tmp/pp_prof; mkdir -p $S
timeout 600 python -m pytest tests/test_transformer_kernels.py tests/test_generator.py tests/test_e2e.py -m gpu -q -x 2>&1 | tail -3
timeout 200 rocprofv3 --kernel-trace --stats -d $S -o trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
python tools/rocpd_kernel_stats.py $S/trace_results.db $O/kernel_stats.md > /dev/null; grep -E "window_attention|instnorm" $O/kernel_stats.md | cut -c1-170
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
