#!/bin/bash
# r06 GPU call 13: the library rebuilt with -fno-slp-vectorize: deform_cols next to convolutions (diag tools), lanes reproducibility incl.
# the feature-propagation lanes, whole GPU suite, bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call13; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
python tools/diag_deform_neighbours.py 2>&1 | grep -v amdgpu | grep "wrong in" | tee $O/deform_neighbours.log | cut -c1-160
DIAG_NEIGHBOUR=step,off6,dcn python tools/diag_featprop_race.py 2>&1 | grep -v amdgpu | grep "neighbour =\|next to" | tee $O/featprop_race.log | cut -c1-260
DIAG_T=80 DIAG_HW=360,640 DIAG_NL=10 python tools/diag_cfg5_repro.py 2>&1 | grep -v "amdgpu\|WARNING" | tail -12 | tee $O/lanes_repro_cfg2.log
DIAG_FEATPROP_ONLY=1 DIAG_T=160 python tools/diag_cfg5_repro.py 2>&1 | grep -v "amdgpu\|WARNING" | tail -10 | tee $O/lanes_repro_cfg5.log
timeout 1500 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('node_call_frames_per_s'), d.get('host_enqueue_ms'), {k: d['parity'][k] for k in ('psnr_db','max_lsb','max_abs_float')}, d['roofline']['frac'], d['roofline']['other']['f16']['frac'], d['roofline']['other']['direct'])
PY
