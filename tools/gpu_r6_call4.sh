#!/bin/bash
# r06 GPU call 4: whole GPU suite at HEAD (persistent patch kernel, transformer + RAFT-update hipGraphs), bench line, and the
# bench line without the two new graphs (host_enqueue_ms A/B).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call4; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 1500 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
PP_GRAPHS_RAFT=0 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_noraftgraph.json 2> $O/bench_noraftgraph.err
for f in bench bench_noraftgraph; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('node_call_frames_per_s'), d.get('host_enqueue_ms'), d['parity']['psnr_db'], d['parity']['max_lsb'], d['roofline']['frac'])
PY
done
