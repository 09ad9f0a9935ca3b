#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 600 python -m pytest tests/test_baseline_configs.py tests/test_e2e.py tests/test_edge_cases.py -q -m gpu -k "cfg4 or chunked or sv or e2e" 2>&1 | tail -2
for ov in 1 0; do
  echo "== PP_SUBVIDEO_OVERLAP=$ov"
  PP_SUBVIDEO_OVERLAP=$ov timeout 300 python tools/run_config.py --config 4 --reps 3 2>&1 | grep -v "done$" | tail -2 | cut -c1-330
  PP_SUBVIDEO_OVERLAP=$ov timeout 300 python tools/run_config.py --config 5 --reps 3 2>&1 | grep -v "done$" | tail -2 | cut -c1-330
done
