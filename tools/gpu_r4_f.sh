#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python -m pytest tests/test_raft_kernels.py tests/test_raft.py -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);print(b['value'], b['ms_per_step'], b['roofline']['corr_lookup'], b['parity']['psnr_db'], b['parity']['flow_max_px'])"
