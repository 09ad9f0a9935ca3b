#!/bin/bash
# r06 GPU call 11: whole GPU suite at HEAD (stream lanes for the windows, RAFT's directions / encoders, feature propagation), then
# the rocprofv3 profile of bench.py (kernel trace + FETCH_SIZE / WRITE_SIZE passes -> profiles/r06_*), then the default bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call11; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 1500 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
bash tools/profile_bench.sh r06 > $O/profile.log 2>&1; tail -3 $O/profile.log | cut -c1-300
