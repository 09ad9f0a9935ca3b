#!/bin/bash
# GPU call (round 2, C): GPU suite, conv family A/B (halo f16 / split-K / conflict-free swizzle), bench stage times, LDS counters.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2c; mkdir -p $O; S=/tmp/pp_pmc; mkdir -p $S
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
grep -E "^cfg|geometry:|e2e_|FAILED|Error" $O/pytest_gpu.log | cut -c1-330
timeout 60 tools/convbench > $O/cb_default.json 2>&1
PP_CONV_HALO=0 timeout 60 tools/convbench > $O/cb_nohalo.json 2>&1
PP_CONV_KSPLIT=0 timeout 60 tools/convbench rfc_step_f16 rfc_off0_f16 rfc_bb2_f16 rfc_dcn_f16 > $O/cb_noksplit.json 2>&1
echo "== default | no halo"; paste -d' ' $O/cb_default.json $O/cb_nohalo.json | cut -c1-220
echo "== no ksplit"; cat $O/cb_noksplit.json
PP_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench.log 2>&1; grep -E "stage ms" $O/bench.log | tail -1; tail -1 $O/bench.log | cut -c1-1200
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 60 rocprofv3 --pmc $C --kernel-trace -d $S -o p$i -- tools/convbench raft_convc2_f32x2 raft_gru_1x5_f32x2 enc_3x3_256_384_f16 rfc_off0_f16 > $O/p$i.log 2>&1
done
python tools/rocpd_pmc_multi.py $O/pmc.md $S/p*_results.db | cut -c1-420
