#!/bin/bash
# r06 final GPU call: what the driver runs at round end -- the whole GPU suite, smoke(), bench.py with its default arguments --
# plus the rocprofv3 passes of tools/profile_bench.sh.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_final; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep -v amdgpu | tee $O/smoke.log | tail -2
timeout 900 python bench.py > $O/bench_default_args.json 2> $O/bench_default_args.err; tail -c 400 $O/bench_default_args.json
bash tools/profile_bench.sh r06 2>&1 | tail -4 | cut -c1-300
