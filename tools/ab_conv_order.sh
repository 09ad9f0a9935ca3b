#!/bin/bash
# A/B of the flat-tile kernels' tile order on the MI355X (gpurun): XCD-contiguous, channel-adjacent (default) against launch
# order (PP_CONV_ORDER=launch, the order of rounds 1-3).  Per-layer times, the bench line and the HBM traffic both ways.
#   gpurun --timeout 900 -- 'bash tools/ab_conv_order.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab_conv_order; mkdir -p $O; S=/tmp/pp_ab; mkdir -p $S
for order in default launch; do
  if [ $order = launch ]; then export PP_CONV_ORDER=launch; else unset PP_CONV_ORDER; fi
  timeout 150 python tools/profile_layers.py > $O/layers_$order.log 2>&1
  timeout 150 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_$order.json 2> $O/bench_$order.err
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $S -o fetch_$order -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/fetch_$order.log 2>&1
  python tools/rocpd_pmc_stats.py $S/fetch_${order}_results.db FETCH_SIZE $O/pmc_fetch_$order.json > /dev/null
  echo "== $order"; head -30 $O/layers_$order.log | cut -c1-110; python -c "import json;b=json.load(open('$O/bench_$order.json'));print(b['value'], b['ms_per_step'], b['roofline']['other']['f16'])"
done
