#!/bin/bash
# direct kernel with the fp32 weight table (scalar-cache weights) vs LDS-decoded weights; GPU suite; bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2u; mkdir -p $O
S="raft_fh2_f32x2 dec6_f16 rfc_up2_f16"
echo "== table"; timeout 60 tools/convbench $S 2>&1 | tee $O/convbench_table.log
echo "== LDS weights (PP_CONV_DIRECT_TABLE=0)"; PP_CONV_DIRECT_TABLE=0 timeout 60 tools/convbench $S 2>&1 | tee $O/convbench_lds.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2u/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['other'], d['node_call']['ms'], d['f32_exact']['value'])
PY
