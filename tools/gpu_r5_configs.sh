#!/bin/bash
# r05: BASELINE configs[2], [3], [4] (640 frames on one GPU; + parity against cfg4_640f_node) and [5] through tools/run_config.py
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_configs; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
for c in 2 3 4 5; do echo "## cfg $c"; PP_TIMING=1 timeout 300 python tools/run_config.py --config $c --reps 3 2>&1 | grep -v "amdgpu.ids\|^$" | tail -4; done | tee $O/configs.log
echo "## cfg 4 at its stated length against the reference's own 640-frame output" | tee -a $O/configs.log
timeout 300 python tools/run_config.py --parity cfg4_640f_node --reps 1 2>&1 | grep -v "amdgpu.ids\|^$" | tail -3 | cut -c1-900 | tee -a $O/configs.log
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys;b=json.loads(sys.stdin.read());print('bench', b['value'], b['ms_per_step'], 'enqueue', b['host_enqueue_ms'], b['conv_param_cache_of_that_step'], b['roofline']['frac'])" | tee -a $O/configs.log
