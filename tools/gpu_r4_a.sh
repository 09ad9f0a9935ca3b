#!/bin/bash
# r04 call A: A/B of the flat-tile order (per-layer times, bench line, FETCH_SIZE both ways) + a kernel trace at HEAD.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 800 bash tools/ab_conv_order.sh > gpurun_out/ab_conv_order.log 2>&1; tail -70 gpurun_out/ab_conv_order.log | cut -c1-150
O=gpurun_out/prof_r04a; mkdir -p $O; S=/tmp/pp_prof; mkdir -p $S
timeout 200 rocprofv3 --kernel-trace --stats -d $S -o trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
python tools/rocpd_kernel_stats.py $S/trace_results.db $O/kernel_stats.md > /dev/null
head -40 $O/kernel_stats.md | cut -c1-180
