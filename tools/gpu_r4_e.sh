#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in device overlap device overlap; do
  PP_OUTPUT=$mode timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);n=b['node_call'];print('$mode', 'pipeline', b['ms_per_step'], 'node_call', n['ms'], n['frames_per_s'], n['breakdown_ms'])"
done
