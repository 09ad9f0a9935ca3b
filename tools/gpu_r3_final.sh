#!/bin/bash
# Final GPU call of round 3: profiles at HEAD (tools/profile_bench.sh), BASELINE configs 3 / 5, attention counters, per-layer conv table.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_bench.sh r03
O=gpurun_out/prof_r03
for c in 3 5; do timeout 200 python tools/run_config.py --config $c --reps 2 2>&1 | grep -E "^\{|stage ms" | tail -2; done | tee $O/cfg3_cfg5.log
bash tools/pmc_attention.sh r03 > /dev/null 2>&1; cp gpurun_out/attn_r03/pmc.md $O/attention_counters.md; head -30 $O/attention_counters.md
timeout 150 python tools/profile_layers.py > $O/conv_layers.log 2>&1; tail -5 $O/conv_layers.log | cut -c1-200
