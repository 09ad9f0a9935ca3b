#!/bin/bash
# Run on the MI355X box (via gpurun): rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes of bench.py,
# summarised on the box (the rocpd databases are too large to travel back), then a clean bench run.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_final; mkdir -p $O; S=/tmp/pp_prof; mkdir -p $S
timeout 240 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 150 rocprofv3 --kernel-trace --stats -d $S -o trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
python tools/rocpd_kernel_stats.py $S/trace_results.db $O/kernel_stats.md > /dev/null
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $S -o fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/fetch.log 2>&1
python tools/rocpd_pmc_stats.py $S/fetch_results.db FETCH_SIZE $O/pmc_fetch_size.json > /dev/null
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $S -o write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/write.log 2>&1
python tools/rocpd_pmc_stats.py $S/write_results.db WRITE_SIZE $O/pmc_write_size.json > /dev/null
python tools/make_traffic.py $O/pmc_fetch_size.json $O/pmc_write_size.json $O/traffic.json conv_split_kernel=f32x2 conv_igemm_kernelIDF16_=f16 'conv_igemm_kernel<float'=f32 > /dev/null
cp $O/traffic.json profiles/r01_traffic.json   # bench.py reads it for roofline.traffic
timeout 200 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
head -12 $O/kernel_stats.md; tail -c 400 $O/bench.json
