#!/bin/bash
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
# the whole step with the variant swapped in, only if the kernel A/B says it is worth the minute
if [ "${R5_FULL:-0}" = 1 ]; then bash tools/ab_variant.sh trim64 > $O/ab_trim64_bench.log 2>&1; tail -12 $O/ab_trim64_bench.log; fi
