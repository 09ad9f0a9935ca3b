#!/bin/bash
# r05: the whole GPU suite (incl. the long-clip fixtures) with its log kept for profiles/, then the rocprof passes + bench of
# tools/profile_bench.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_suite; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids" > $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
bash tools/profile_bench.sh r05 2>&1 | tail -20
