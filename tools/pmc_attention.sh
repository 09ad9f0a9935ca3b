#!/bin/bash
# SQ / LDS / L2 counters of the window-attention kernel at the cfg-2 geometry (rocprofv3 --pmc passes, one group per pass).
#   gpurun --timeout 600 -- 'bash tools/pmc_attention.sh r03'
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/attn_$TAG; mkdir -p $O; S=/tmp/pp_pmc_attn; mkdir -p $S
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --kernel-trace -d $S -o p$i -- python tools/bench_attention.py --reps 3 > $O/p$i.log 2>&1
done
python - "$O/pmc.md" $S/p*_results.db <<'PY'
import sqlite3, sys
out, paths = sys.argv[1], sys.argv[2:]
agg = {}
for path in paths:
    db = sqlite3.connect(path)
    for name, ctr, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if "window_attention" not in name:
            continue
        a = agg.setdefault(ctr, [0, 0.0]); a[0] += 1; a[1] += v
lines = ["| counter | per launch |", "|---|---|"] + [f"| {c} | {a[1] / a[0]:.4g} |" for c, a in sorted(agg.items())]
open(out, "w").write("\n".join(lines) + "\n"); print("\n".join(lines))
PY
