#!/bin/bash
# r06 GPU call 22: paired-quad (16-byte) stores in the GEMM kernel: per-layer A/B, GEMM tests, whole-clip A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call22; mkdir -p $O
python tools/bench_gemm_oct.py 2>&1 | grep -v amdgpu | tee $O/bench_gemm_oct.log
timeout 600 python -m pytest tests/test_conv.py tests/test_transformer_kernels.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
  PP_CONV_EPI_OCT=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_oct${v}_$rep.json 2>/dev/null
  python - $O/bench_oct${v}_$rep.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("PP_CONV_EPI_OCT=%s: %.2f frames/s, %.1f ms, f16 frac %.4f, parity max_lsb %s psnr %.2f" % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['other']['f16']['frac'], d['parity']['max_lsb'], d['parity']['psnr_db']))
PY
done; done
