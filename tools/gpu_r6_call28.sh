#!/bin/bash
# r06 GPU call 28: ABI v12 many_images (flow completion's batch layers off the split-K kernel): layer microbench, whole GPU suite, bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call28; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
PP_TIMING=1 timeout 400 python tools/run_config.py --config 2 --reps 3 2>&1 | grep "stage ms\|frames_per_s" | tail -2 | cut -c1-400
unset PP_ALLOW_SYNTHETIC_WEIGHTS
timeout 1800 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
grep "cfg3_80f_node\|cfg5_160f_node\|cfg2_80f_node\|cfg4_640f" $O/pytest_gpu.log | cut -c1-330
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2>/dev/null
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%.2f frames/s, %.1f ms, node %.1f, f16 frac %.4f (%.1f ms), f32x2 %.4f, parity %s dB max_lsb %s max_abs_float %s" % (d['value'], d['ms_per_step'], d['node_call_frames_per_s'], d['roofline']['other']['f16']['frac'], d['roofline']['other']['f16']['ms'], d['roofline']['frac'], d['parity']['psnr_db'], d['parity']['max_lsb'], d['parity']['max_abs_float']))
PY
