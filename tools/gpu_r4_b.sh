#!/bin/bash
# r04 call: split-K ring depth A/B (per-layer times + bench line), saturation test, distributed tests
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4_b; mkdir -p $O
for nst in 4 3; do
  export PP_CONV_KSPLIT_NST=$nst
  timeout 200 python tools/profile_layers.py 2>/dev/null | grep "M7200\|total" > $O/layers_nst$nst.log
  timeout 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_nst$nst.json 2>/dev/null
  echo "== NST $nst"; cat $O/layers_nst$nst.log | cut -c1-100; python -c "import json;b=json.load(open('$O/bench_nst$nst.json'));print(b['value'], b['ms_per_step'], b['roofline']['other']['f16'], b['parity']['psnr_db'])"
done
unset PP_CONV_KSPLIT_NST
timeout 300 python -m pytest tests/test_rfc.py tests/test_conv.py -q -m gpu -k "undamped or ksplit or f32_output" 2>&1 | tail -5
