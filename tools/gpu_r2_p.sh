#!/bin/bash
# f16 compile-time 3x3 halo kernel: A/B, parity on the hardware, GPU suite, bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2p; mkdir -p $O
S="enc_3x3_256_384_f16 f16_3x3_256_512 dcn_offset_f16 featprop_bb2_f16 dec_3x3_128_128_f16"
echo "== ct rows of taps"; timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_ct.log
echo "== runtime taps (PP_CONV_HALO_CT=0)"; PP_CONV_HALO_CT=0 timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_rt.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tee $O/pytest_gpu.log | tail -4
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
