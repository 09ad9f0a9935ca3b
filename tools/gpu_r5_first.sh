#!/bin/bash
# (r05 end: the build hooks this script times -- PP_HALO_TRIM64 / PP_HALO_WMID / PP_EPI_STORE_MODE -- were measured and removed from csrc/;
#  TRIM64 is archived in tools/experiments/r05_halo_hooks.patch, the other two are in git history.  Kept as the record of how
#  profiles/r05_ab_*.log and r05_halo_trace_call1_r04_kernel.log were produced.)
# First GPU call of the next round (prepared at the end of r04, when the GPU budget was spent): the occupancy A/B that the r04 SQ
# counters point at (profiles/r04_conv_counters.md: matrix pipe 53-61 % busy at 2 waves per SIMD; one lock-step work-group per CU
# loses, profiles/r04_ab_w8.log).  Variant `trim64` = csrc/conv_halo.hip built with -DPP_HALO_TRIM64: every PP_F32X2 halo layer on
# the 64-channel x (8 x 16) tile (150-155 registers) with the pixel tile trimmed to its halo rows -> THREE independent work-groups
# per CU for 3x3 / 1x5 taps (53.4 / 50.2 KB each; 5x1 stays at two).  Emulator-checked bit-identical to the product
# (PP_EMU_DEFINES=-DPP_HALO_TRIM64).  Build it first, in the build container:
#     bash tools/build_variant.sh trim64 conv_halo -DPP_HALO_TRIM64
# then:  gpurun --timeout 300 -- 'bash tools/gpu_r5_first.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_first; mkdir -p $O
bash tools/ab_convbench.sh "trim64" raft_gru_1x5_f32x2 raft_gru128_1x5_f32x2 raft_gru128_5x1_f32x2 raft_convc2_f32x2 raft_fh1_f32x2 > $O/ab_trim64.log 2>&1
cat $O/ab_trim64.log
# f16 family, no rebuild needed: PP_CONV_HALO_C64=1 = every 3x3 f16 compile-time-tap halo layer on 64-channel tiles (48 KB, 100
# registers: three work-groups per CU instead of two); kernel level, then the whole step with its parity block
for c in 0 1; do echo "== PP_CONV_HALO_C64=$c"; PP_CONV_HALO_C64=$c timeout 40 tools/convbench enc_3x3_256_384_f16 f16_3x3_256_512 featprop_bb2_f16 dec_3x3_128_128_f16; done 2>&1 | tee $O/ab_c64.log
for c in 0 1; do PP_CONV_HALO_C64=$c timeout 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys;b=json.loads(sys.stdin.read());print('C64=$c', b['value'], b['ms_per_step'], b['parity']['psnr_db'], b['parity']['flow_max_px'])"; done 2>&1 | tee -a $O/ab_c64.log
# the whole step with the variant swapped in, only if the kernel A/B says it is worth the minute
if [ "${R5_FULL:-0}" = 1 ]; then bash tools/ab_variant.sh trim64 > $O/ab_trim64_bench.log 2>&1; tail -12 $O/ab_trim64_bench.log; fi
