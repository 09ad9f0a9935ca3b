#!/bin/bash
# end-of-round validation: GPU suite, default bench line (with cpu_baseline / node_call / parity), per-layer conv profile
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2t; mkdir -p $O
timeout 120 tools/convbench fc1_f16 fc2_f16 proj_f16 qkv_f16 2>&1 | tee $O/convbench.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 300 python tools/profile_layers.py > $O/layers.log 2>&1; head -3 $O/layers.log | tail -2
