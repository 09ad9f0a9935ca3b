#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2f; mkdir -p $O
PP_TEST_BACKEND=hip PP_CONV_TILE=large PP_CONV_HALO=force PP_CONV_KSPLIT=0 PYTHONPATH=. timeout 200 python tests/test_conv.py > $O/conv_halo2.log 2>&1; echo "conv halo2 parity rc=$?"; tail -2 $O/conv_halo2.log
for i in 1 2; do
timeout 60 tools/convbench raft_gru_1x5_f32x2 raft_convc2_f32x2 raft_fh1_f32x2 > $O/cb_halo2_$i.json 2>&1
PP_CONV_HALO2=0 timeout 60 tools/convbench raft_gru_1x5_f32x2 raft_convc2_f32x2 raft_fh1_f32x2 > $O/cb_halo1_$i.json 2>&1
echo "== two-group | one-group"; paste -d' ' $O/cb_halo2_$i.json $O/cb_halo1_$i.json | cut -c1-200
done
timeout 300 python -m pytest tests/test_raft.py tests/test_e2e.py -m gpu -q -s 2>&1 | tail -4 | cut -c1-300
PP_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; grep -E "stage ms" $O/bench.log | tail -1; tail -1 $O/bench.log | cut -c1-300
