#!/bin/bash
# streamed node output: node tests, BASELINE-config node tests, bench with the node_call leg (stream vs blocking)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2r; mkdir -p $O
timeout 300 python -m pytest tests/test_nodes.py -m gpu -q -x 2>&1 | tail -2
timeout 500 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2r/bench.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('node_call'))[:600])
PY
PP_OUTPUT=host timeout 500 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_host.log 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2r/bench_host.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('node_call'))[:600])
PY
