#!/bin/bash
# bisect: fp16=disable flow completion at 640x360 and the sharded-vs-single mismatch, under kernel-family / graph toggles
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python -m pytest tests/test_baseline_configs.py -q -s -m gpu -k "cfg2_24f_node and disable" > $O/$name.log 2>&1; echo "$name: $(grep -E '^cfg2' $O/$name.log | cut -c1-200)"; }
run default A=1
run nographs PP_GRAPHS=0
run nohalo PP_CONV_HALO=0
run classic_nohalo PP_CONV_HALO=0 PP_CONV_TILE=classic
run large_nohalo PP_CONV_HALO=0 PP_CONV_TILE=large
run exact PP_F32_GEMM=exact
run exact_nographs PP_F32_GEMM=exact PP_GRAPHS=0
for g in 1 0; do PP_GRAPHS=$g timeout 300 python -m pytest tests/test_distributed.py -q -m gpu -k "bit_identical or two_processes" > $O/dist_g$g.log 2>&1; echo "dist graphs=$g: $(tail -1 $O/dist_g$g.log)"; done
PP_CONV_KSPLIT=0 timeout 300 python -m pytest tests/test_distributed.py -q -m gpu -k "bit_identical or two_processes" > $O/dist_noks.log 2>&1; echo "dist no ksplit: $(tail -1 $O/dist_noks.log)"
