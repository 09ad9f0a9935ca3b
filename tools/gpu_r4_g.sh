#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_conv.py tests/test_transformer_kernels.py tests/test_generator.py -q -m gpu -k "linear_of_unfold or fold or generator" 2>&1 | tail -2
for mode in fused copy fused copy; do
  PP_FC2_UNFOLD=$mode timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);print('$mode', b['value'], b['ms_per_step'], b['roofline']['other']['f16'], b['parity']['psnr_db'], b['parity']['max_lsb'])"
done
