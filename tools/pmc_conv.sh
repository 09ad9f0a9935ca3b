#!/bin/bash
# Run on the MI355X box (via gpurun): SQ counter passes over tools/pmc_conv.py (PMC only + kernel trace).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_conv; mkdir -p $O; S=/tmp/pp_pmc; mkdir -p $S
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_ANY SQ_INST_CYCLES_VALU SQ_INSTS_VALU_CVT SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $C --kernel-trace -d $S -o p$i -- python tools/pmc_conv.py > $O/p$i.log 2>&1
done
python tools/rocpd_pmc_multi.py $O/pmc.md $S/p*_results.db | cut -c1-260
