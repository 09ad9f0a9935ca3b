#!/bin/bash
# BASELINE.json configs[2] (outpaint 768x360) and configs[4] (160 f 1280x720, nl 20) end to end on the MI355X, with kernel stats of cfg 5
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cfgs; mkdir -p $O; S=/tmp/pp_cfg; mkdir -p $S
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 300 python tools/run_config.py --config 3 --reps 3 > $O/cfg3.log 2>&1; grep -E '"config"' $O/cfg3.log | tail -2
timeout 600 python tools/run_config.py --config 5 --reps 2 > $O/cfg5.log 2>&1; grep -E '"config"|stage ms' $O/cfg5.log | tail -3
timeout 400 rocprofv3 --kernel-trace --stats -d $S -o c5 -- python tools/run_config.py --config 5 --reps 1 > $O/cfg5_trace.log 2>&1
python tools/rocpd_kernel_stats.py $S/c5_results.db $O/cfg5_kernel_stats.md > /dev/null; head -12 $O/cfg5_kernel_stats.md | cut -c1-150
