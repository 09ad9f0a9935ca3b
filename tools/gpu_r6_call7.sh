#!/bin/bash
# r06 GPU call 7: the tightened parity asserts over the whole GPU suite, smoke(), the outlier diagnostic with the completed-flow
# differences around each byte beyond 2 LSB, the bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call7; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 1500 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
timeout 600 python tools/diag_lsb_outliers.py --run --case cfg3_80f_node --fp16 enable --lsb 2 > $O/outliers_cfg3_80f.log 2>&1
timeout 900 python tools/diag_lsb_outliers.py --run --case cfg5_160f_node --fp16 enable --lsb 2 > $O/outliers_cfg5_160f.log 2>&1
grep -h "completed-flow\|^frame\|bytes beyond" $O/outliers_cfg3_80f.log $O/outliers_cfg5_160f.log | cut -c1-420
timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('node_call_frames_per_s'), d.get('host_enqueue_ms'), {k: d['parity'][k] for k in ('psnr_db','max_lsb','max_abs_float')}, d['roofline']['frac'])
PY
