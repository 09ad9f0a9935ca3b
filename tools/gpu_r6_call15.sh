#!/bin/bash
# r06 GPU call 15: HEAD (library built with -fno-slp-vectorize, feature-propagation lanes on): whole GPU suite, smoke(), the
# rocprofv3 passes + bench line (tools/profile_bench.sh), configs 2-5.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call15; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep -v amdgpu | tee $O/smoke.log | tail -2
bash tools/profile_bench.sh r06 2>&1 | tail -20
bash tools/gpu_r6_configs.sh 2>&1 | tail -24
