#!/bin/bash
# r06 GPU call 30: f16 halo tiles for the 17..32-channel layers too (PP_CONV_HALO_MINCOUT=17): conv tests under it, whole-clip A/B
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call30; mkdir -p $O
PP_CONV_HALO_MINCOUT=17 timeout 900 python -m pytest tests/test_conv.py tests/test_rfc.py tests/test_generator.py tests/test_e2e.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in 33 17; do
  PP_CONV_HALO_MINCOUT=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_mc${v}_$rep.json 2>/dev/null
  python - $O/bench_mc${v}_$rep.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("PP_CONV_HALO_MINCOUT=%s: %.2f frames/s, %.1f ms, f16 %.4f, parity max_lsb %s psnr %.2f max_abs_float %s" % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['other']['f16']['frac'], d['parity']['max_lsb'], d['parity']['psnr_db'], d['parity']['max_abs_float']))
PY
done; done
