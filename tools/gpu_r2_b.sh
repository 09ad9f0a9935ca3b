#!/bin/bash
# GPU call (round 2, B): whole GPU suite (no -x), SQ / LDS / cache counters of the halo conv kernel, stage timing with hipGraphs.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O; S=/tmp/pp_pmc; mkdir -p $S
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
grep -E "^cfg|geometry:|e2e_|FAILED|Error" $O/pytest_gpu.log | cut -c1-330
rocprofv3 -L > $O/counters_list.txt 2>&1
i=0
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 60 rocprofv3 --pmc $C --kernel-trace -d $S -o p$i -- tools/convbench raft_convc2_f32x2 raft_gru_1x5_f32x2 enc_3x3_256_384_f16 > $O/p$i.log 2>&1
  PP_CONV_HALO=0 timeout 60 rocprofv3 --pmc $C --kernel-trace -d $S -o f$i -- tools/convbench raft_convc2_f32x2 > $O/f$i.log 2>&1
done
python tools/rocpd_pmc_multi.py $O/pmc.md $S/p*_results.db $S/f*_results.db | cut -c1-400
PP_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_graphs.log 2>&1; grep -E "stage ms" $O/bench_graphs.log | tail -2; tail -1 $O/bench_graphs.log | cut -c1-300
PP_GRAPHS=0 PP_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_nographs.log 2>&1; grep -E "stage ms" $O/bench_nographs.log | tail -1; tail -1 $O/bench_nographs.log | cut -c1-300
