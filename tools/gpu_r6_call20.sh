#!/bin/bash
# r06 GPU call 20: A/B of PP_FEATPROP_PIPE (second feature-propagation half next to the first windows) + the lanes regression test
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call20; mkdir -p $O
timeout 600 python -m pytest tests/test_e2e.py -x -q -m gpu -k "lanes" 2>&1 | tail -2
for rep in 1 2; do for v in 0 1; do
  PP_FEATPROP_PIPE=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_pipe${v}_$rep.json 2>/dev/null
  python - $O/bench_pipe${v}_$rep.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("PP_FEATPROP_PIPE=%s: %.2f frames/s, %.1f ms, parity max_lsb %s" % (sys.argv[2], d['value'], d['ms_per_step'], d['parity']['max_lsb']))
PY
done; done
