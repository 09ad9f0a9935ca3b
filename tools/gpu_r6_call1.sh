#!/bin/bash
# r06 GPU call 1: (1) where the bytes beyond 2 LSB are (cfg3_80f / cfg5_160f, fp16 enable) with OUR float values per window
# (tools/diag_lsb_outliers.py --run), (2) SQ counters of conv_gemm_f16_kernel / conv_ksplit_kernel, (3) the bench line of HEAD.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call1; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 600 python tools/diag_lsb_outliers.py --run --case cfg3_80f_node --fp16 enable --lsb 2 > $O/outliers_cfg3_80f.log 2>&1
timeout 900 python tools/diag_lsb_outliers.py --run --case cfg5_160f_node --fp16 enable --lsb 2 > $O/outliers_cfg5_160f.log 2>&1
tail -n 12 $O/outliers_cfg3_80f.log $O/outliers_cfg5_160f.log | cut -c1-400
PMC_CASES=tf_,rfc_ PMC_PASSES="1 4 5" bash tools/pmc_conv.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc_conv/pmc.md $O/pmc_gemm_ksplit.md
timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
