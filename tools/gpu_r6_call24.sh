#!/bin/bash
# r06 GPU call 24: 2 x 2-block upsample kernel: microbench both forms, kernel tests, whole-clip A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call24; mkdir -p $O
for v in 0 1; do echo "PP_UPSAMPLE_B4=$v"; PP_UPSAMPLE_B4=$v python tools/bench_upsample.py 2>&1 | grep -v amdgpu; done | tee $O/bench_upsample.log
timeout 900 python -m pytest tests/test_sample_kernels.py tests/test_conv.py tests/test_generator.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
  PP_UPSAMPLE_B4=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_b4${v}_$rep.json 2>/dev/null
  python - $O/bench_b4${v}_$rep.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("PP_UPSAMPLE_B4=%s: %.2f frames/s, %.1f ms, parity max_lsb %s psnr %.2f max_abs_float %s" % (sys.argv[2], d['value'], d['ms_per_step'], d['parity']['max_lsb'], d['parity']['psnr_db'], d['parity']['max_abs_float']))
PY
done; done
