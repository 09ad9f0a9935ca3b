#!/bin/bash
# A/B: <= 4-output-channel 3x3 f16 layers on 16-channel halo MFMA tiles (PP_CONV_SMALL_HALO=1) vs the vector-ALU kernel
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1 0 1; do
  PP_CONV_SMALL_HALO=$m timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);r=b['roofline'];print('small_halo=$m', b['value'], b['ms_per_step'], r['other'], b['parity']['psnr_db'], b['parity']['max_lsb'])"
done
PP_CONV_SMALL_HALO=1 timeout 200 python tools/profile_layers.py 2>/dev/null | grep "cout3 \|cout2 " | cut -c1-100
