#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2h; mkdir -p $O; S=/tmp/pp_prof; mkdir -p $S
timeout 300 python -m pytest tests/test_transformer_kernels.py tests/test_generator.py tests/test_e2e.py -m gpu -q 2>&1 | tail -3
timeout 200 rocprofv3 --kernel-trace --stats -d $S -o trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
python tools/rocpd_kernel_stats.py $S/trace_results.db $O/kernel_stats.md > /dev/null; head -30 $O/kernel_stats.md | cut -c1-170
PP_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; grep -E "stage ms" $O/bench.log | tail -1; tail -1 $O/bench.log | cut -c1-300
