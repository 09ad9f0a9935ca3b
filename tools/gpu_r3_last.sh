#!/bin/bash
# Last GPU call of round 3 (gpurun): A/B of the one-launch deformable convolution, its parity tests, the edge-case node
# fixtures and the convolution suite after the library prune.  Everything under its own timeout; logs under gpurun_out/r3_last.
#   gpurun --timeout 240 -- 'bash tools/gpu_r3_last.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_last; mkdir -p $O
timeout 60 python tools/bench_deform.py > $O/bench_deform.jsonl 2> $O/bench_deform.err; cut -c1-210 $O/bench_deform.jsonl; tail -2 $O/bench_deform.err
timeout 90 python -m pytest tests/test_sample_kernels.py tests/test_e2e.py -m gpu -q -s > $O/pytest_deform.log 2>&1; tail -3 $O/pytest_deform.log | cut -c1-200
PP_DEFORM_FUSED=1 timeout 150 python -m pytest tests/test_edge_cases.py -m gpu -q -s --maxfail=8 > $O/pytest_edge.log 2>&1; tail -3 $O/pytest_edge.log | cut -c1-200
grep "fp16=" $O/pytest_edge.log | cut -c1-60,150-260
timeout 60 python -m pytest tests/test_conv.py -m gpu -q > $O/pytest_conv.log 2>&1; tail -2 $O/pytest_conv.log
