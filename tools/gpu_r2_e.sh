#!/bin/bash
# GPU call (round 2, E): whole GPU suite after the fixes + bench with all extras
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log | cut -c1-300
grep -E "^cfg|geometry:|e2e_|FAILED" $O/pytest_gpu.log | cut -c1-330
PP_TIMING=1 timeout 500 python bench.py --steps 5 --warmup 2 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-3500
