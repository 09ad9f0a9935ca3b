#!/bin/bash
# r05 GPU call 2: the LDS-transposed epilogue (PP_CONV_EPI default) against the direct form (PP_CONV_EPI=direct), kernel level (C++
# client, outputs hashed: must be bit-identical) and whole step; GPU conv tests; host_enqueue_ms without PP_TIMING.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_call2; mkdir -p $O
export PP_CONVBENCH_SUM=1
for e in direct lds; do echo "== PP_CONV_EPI=$e"; PP_CONV_EPI=$e timeout 120 tools/convbench raft_gru_1x5_f32x2 raft_gru128_5x1_f32x2 raft_convc2_f32x2 raft_fh1_f32x2 enc_3x3_256_384_f16 f16_3x3_256_512 dcn_offset_f16 fc1_f16 qkv_f16 fc2_f16 proj_f16 rfc_step_f16 rfc_off0_f16 rfc_bb2_f16 rfc_dcn_f16 featprop_bb2_f16 dec_3x3_128_128_f16; done 2>&1 | tee $O/ab_epi.log
timeout 600 python -m pytest tests/test_conv.py tests/test_sample_kernels.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_conv.log
for e in direct lds; do PP_CONV_EPI=$e timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>$O/bench_$e.err | grep "^{" | tail -1 > $O/bench_$e.json; python -c "import json,sys;b=json.load(open('$O/bench_$e.json'));print('EPI=$e', b['value'], b['ms_per_step'], 'enqueue', b['host_enqueue_ms'], b['roofline']['frac'], b['roofline']['other'], b['parity']['psnr_db'], b['parity']['max_lsb'], b['parity']['flow_max_px'])"; done 2>&1 | tee $O/bench_ab.log
