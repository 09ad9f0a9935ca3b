#!/bin/bash
# r06: BASELINE configs[2], [3], [4] (640 frames on one GPU) and [5] through tools/run_config.py (node call, models cached, best of 3)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_configs; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
for c in 2 3 4 5; do echo "## cfg $c"; PP_TIMING=1 timeout 400 python tools/run_config.py --config $c --reps 3 2>&1 | grep -v "amdgpu.ids\|^$" | tail -4 | cut -c1-700; done | tee $O/configs.log
