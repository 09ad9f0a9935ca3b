#!/bin/bash
# (r05 end: the build hooks this script times -- PP_HALO_TRIM64 / PP_HALO_WMID / PP_EPI_STORE_MODE -- were measured and removed from csrc/;
#  TRIM64 is archived in tools/experiments/r05_halo_hooks.patch, the other two are in git history.  Kept as the record of how
#  profiles/r05_ab_*.log and r05_halo_trace_call1_r04_kernel.log were produced.)
# r05 GPU call 1: (1) does the matrix pipe honour f16 subnormal inputs (tools/probes/mfma_denorm); (2) the occupancy A/B prepared at
# the end of r04 (trim64 = three work-groups per CU) and the mid-burst weight copy (wmid) against the product, C++ client, outputs
# hashed; (3) the s_memtime phase trace of the PP_F32X2 halo kernels (tools/variants/trace.so), two work-groups per CU and solo;
# (4) PP_CONV_HALO_C64 for the f16 family; (5) one short bench line (host_enqueue_ms).
# Build first:  for v in "trace -DPP_HALO_TRACE" "wmid -DPP_HALO_WMID" "trim64 -DPP_HALO_TRIM64"; do bash tools/build_variant.sh ${v%% *} conv_halo ${v#* }; done
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call1; mkdir -p $O
L=comfyui_propainter_nodes_amd/libpropainter_mi355.so
tools/probes/mfma_denorm 2>&1 | tee $O/mfma_denorm.log
SHAPES="raft_gru_1x5_f32x2 raft_gru128_1x5_f32x2 raft_gru128_5x1_f32x2 raft_convc2_f32x2 raft_fh1_f32x2"
bash tools/ab_convbench.sh "trim64 wmid" $SHAPES > $O/ab_trim64_wmid.log 2>&1; cat $O/ab_trim64_wmid.log
cp $L /tmp/product.so; cp tools/variants/trace.so $L
for solo in 0 1; do PP_HALO_TRACE_SOLO=$solo timeout 60 tools/convbench $SHAPES 2>&1 | grep -v "^$"; done | tee $O/halo_trace.log
cp /tmp/product.so $L
for c in 0 1; do echo "== PP_CONV_HALO_C64=$c"; PP_CONV_HALO_C64=$c timeout 40 tools/convbench enc_3x3_256_384_f16 f16_3x3_256_512 featprop_bb2_f16 dec_3x3_128_128_f16; done 2>&1 | tee $O/ab_c64.log
for c in 0 1; do PP_CONV_HALO_C64=$c PP_TIMING=1 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>$O/bench_c64_$c.err | grep "^{" | tail -1 > $O/bench_c64_$c.json; python -c "import json,sys;b=json.load(open('$O/bench_c64_$c.json'));print('C64=$c', b['value'], b['ms_per_step'], 'enqueue', b['host_enqueue_ms'], b['roofline']['frac'], b['parity']['psnr_db'], b['parity']['flow_max_px'])"; done 2>&1 | tee -a $O/ab_c64.log
