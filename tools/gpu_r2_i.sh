#!/bin/bash
# compile-time-tap halo kernel: A/B against the runtime-tap kernel, hardware parity, then the suite and the bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2i; mkdir -p $O
S="raft_gru_1x5_f32x2 raft_convc2_f32x2 raft_fh1_f32x2"
echo "== ct taps (default)"; timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_ct.log
echo "== runtime taps (PP_CONV_HALO_CT=0)"; PP_CONV_HALO_CT=0 timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_rt.log
timeout 600 python -m pytest tests/test_conv.py -m gpu -q 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tee $O/pytest_gpu.log | tail -5
PP_TIMING=1 timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1; grep -E "stage ms" $O/bench.log | tail -1; tail -1 $O/bench.log | cut -c1-600
PP_CONV_HALO_CT=0 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_rt.log 2>&1; tail -1 $O/bench_rt.log | cut -c1-200
