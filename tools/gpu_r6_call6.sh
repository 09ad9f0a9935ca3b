#!/bin/bash
# r06 GPU call 6: RAFT's iteration body walked over chunks of n pair-directions (PP_RAFT_PCHUNK: producer -> consumer through the
# 256 MB Infinity Cache instead of HBM): bench line per n, and the RAFT tests with a chunked body.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call6; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
PP_RAFT_PCHUNK=20 timeout 600 python -m pytest tests/test_raft.py -x -q -m gpu 2>&1 | tail -2
for n in 0 16 32 48 80; do
  PP_RAFT_PCHUNK=$n timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$n.json 2> $O/bench_$n.err
  python - $O/bench_$n.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("PCHUNK", sys.argv[2], d['value'], d['ms_per_step'], d.get('host_enqueue_ms'), d['parity']['psnr_db'], d['parity']['max_lsb'], d['parity']['flow_max_px'], d['roofline']['frac'])
PY
done
