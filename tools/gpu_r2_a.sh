#!/bin/bash
# GPU call (round 2, A): full GPU suite, halo-vs-flat conv A/B, bench with the new fields.
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
grep -E "^cfg|geometry:|e2e_" $O/pytest_gpu.log | cut -c1-300
timeout 30 tools/convbench > $O/convbench_halo.json 2>&1; PP_CONV_HALO=0 timeout 30 tools/convbench > $O/convbench_flat.json 2>&1
paste -d' ' $O/convbench_flat.json $O/convbench_halo.json | cut -c1-220
PP_TIMING=1 timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; grep -E "stage ms|node ms" $O/bench.log | tail -3; tail -1 $O/bench.log | cut -c1-3000
PP_CONV_HALO=0 PP_TIMING=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_flat.log 2>&1; grep -E "stage ms" $O/bench_flat.log | tail -1; tail -1 $O/bench_flat.log | cut -c1-400
PP_CONV_TILE=classic PP_TIMING=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_classic.log 2>&1; grep -E "stage ms" $O/bench_classic.log | tail -1; tail -1 $O/bench_classic.log | cut -c1-400
