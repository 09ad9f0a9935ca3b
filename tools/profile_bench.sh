#!/bin/bash
# Run on the MI355X box (via gpurun): rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes of bench.py,
# summarised on the box (the rocpd databases are too large to travel back), then a clean bench run.
#   gpurun --timeout 1500 -- 'bash tools/profile_bench.sh r03'
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG; mkdir -p $O; S=/tmp/pp_prof; mkdir -p $S
# r06: the rocprofv3 passes run the stages one launch at a time (no second stream next to the windows / RAFT's directions / the
# encoders / the feature-propagation halves), so that a kernel's duration and counters are its own; the bench line at the end runs
# the default schedule
SERIAL="PP_WINDOW_LANES=1 PP_RAFT_LANES=1 PP_ENC_LANES=1 PP_FEATPROP_LANES=1"
timeout 200 env $SERIAL rocprofv3 --kernel-trace --stats -d $S -o trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
python tools/rocpd_kernel_stats.py $S/trace_results.db $O/kernel_stats.md > /dev/null
timeout 150 env $SERIAL rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $S -o fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
python tools/rocpd_pmc_stats.py $S/fetch_results.db FETCH_SIZE $O/pmc_fetch_size.json > /dev/null
timeout 150 env $SERIAL rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $S -o write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/write.log 2>&1
python tools/rocpd_pmc_stats.py $S/write_results.db WRITE_SIZE $O/pmc_write_size.json > /dev/null
python tools/make_traffic.py $O/pmc_fetch_size.json $O/pmc_write_size.json $O/traffic.json conv_split_kernel=f32x2 conv_halo_split_ct_kernel=f32x2 \
  conv_igemm_kernelIDF16_=f16 conv_halo_f16_kernel=f16 conv_halo_f16_ct_kernel=f16 conv_ksplit_kernel=f16 'conv_igemm_kernel<float'=f32 \
  window_attention_f16_kernel=attention corr_lookup_kernel=corr_lookup conv_gemm_f16_kernel=f16 conv_patch_split_kernel=f32x2 > /dev/null
cp $O/traffic.json profiles/${TAG}_traffic.json   # bench.py reads the newest profiles/r*_traffic.json for roofline.traffic
timeout 400 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
head -14 $O/kernel_stats.md | cut -c1-160; tail -c 700 $O/bench.json
