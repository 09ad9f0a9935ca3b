#!/bin/bash
# direct small-Cout kernel: A/B, parity on the hardware, e2e tests, bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2n; mkdir -p $O
S="raft_fh2_f32x2 dec6_f16 rfc_up2_f16"
echo "== direct"; timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_direct.log
echo "== MFMA tiles (PP_CONV_DIRECT=0)"; PP_CONV_DIRECT=0 timeout 120 tools/convbench $S 2>&1 | tee $O/convbench_mfma.log
timeout 900 python -m pytest tests/test_conv.py tests/test_raft.py tests/test_e2e.py tests/test_baseline_configs.py -m gpu -q -x 2>&1 | tail -5
PP_TIMING=1 timeout 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; grep -E "stage ms" $O/bench.log | tail -1; tail -1 $O/bench.log | cut -c1-300
