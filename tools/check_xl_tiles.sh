#!/bin/bash
# Run on the MI355X box (via gpurun): parity of the experimental 8-wave 256x128 conv tiles, then their effect end to end.
#   gpurun --timeout 400 -- 'bash tools/check_xl_tiles.sh'
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/xl; mkdir -p $O
# 1. every conv test case on the hardware with the 8-wave tiles forced (the child of test_conv.py reads the env)
PP_TEST_BACKEND=hip PP_CONV_TILE=xlforce PYTHONPATH=. timeout 100 python tests/test_conv.py > $O/conv_xlforce.log 2>&1; echo "conv xlforce rc=$?"
PP_TEST_BACKEND=hip PP_CONV_TILE=tiny PYTHONPATH=. timeout 100 python tests/test_conv.py > $O/conv_tiny.log 2>&1; echo "conv tiny rc=$?"
# 2. stage + end-to-end parity with the size rule
PP_CONV_TILE=xl timeout 200 python -m pytest tests/test_raft.py tests/test_rfc.py tests/test_generator.py tests/test_e2e.py -m gpu -x -q > $O/e2e_xl.log 2>&1; tail -2 $O/e2e_xl.log
# 3. micro-benchmark and clip throughput, default vs xl
timeout 20 tools/convbench > $O/convbench_default.json; PP_CONV_TILE=xl timeout 20 tools/convbench > $O/convbench_xl.json
paste -d' ' $O/convbench_default.json $O/convbench_xl.json | cut -c1-200
for mode in default xl tiny; do
  if [ $mode = default ]; then unset PP_CONV_TILE; else export PP_CONV_TILE=$mode; fi
  PP_TIMING=1 timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "stage ms|\"value\"" | tail -2 | cut -c1-260
done
