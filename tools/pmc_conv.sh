#!/bin/bash
# Run on the MI355X box (via gpurun): SQ / cache counter passes over tools/pmc_conv.py (PMC only + kernel trace).
# PMC_PASSES="1 4" restricts the run to those passes (r04: the SQ pair, ~20 s each).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_conv; mkdir -p $O; S=/tmp/pp_pmc; mkdir -p $S
PASSES=" ${PMC_PASSES:-1 2 3 4 5} "
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  case "$PASSES" in *" $i "*) ;; *) continue ;; esac
  timeout 60 rocprofv3 --pmc $C --kernel-trace -d $S -o p$i -- python tools/pmc_conv.py > $O/p$i.log 2>&1
done
python tools/rocpd_pmc_multi.py $O/pmc.md $S/p*_results.db | cut -c1-260
