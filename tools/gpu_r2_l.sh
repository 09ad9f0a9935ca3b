#!/bin/bash
# SQ counters: 16-row one-wave-per-SIMD kernel vs the 8-row kernels on the same shapes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2l; mkdir -p $O; S=/tmp/pp_pmc; mkdir -p $S
for V in force 0; do
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  PP_CONV_HALO_TALL=$V timeout 60 rocprofv3 --pmc $C --kernel-trace -d $S -o t${V}_p$i -- tools/convbench raft_convc2_f32x2 raft_fh1_f32x2 > $O/t${V}_p$i.log 2>&1
done
done
python tools/rocpd_pmc_multi.py $O/pmc.md $S/t*_results.db | cut -c1-420
