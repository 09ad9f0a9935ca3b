#!/bin/bash
# r05 GPU call 3: the compact (variant-dispatched, LDS-transposed) epilogue: kernel level (hashes must equal call 2's), the phase
# trace's epilogue figure, GPU conv tests, whole step.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R5_OUT:-r5_call3}; mkdir -p $O
L=comfyui_propainter_nodes_amd/libpropainter_mi355.so
export PP_CONVBENCH_SUM=1
timeout 120 tools/convbench raft_gru_1x5_f32x2 raft_gru128_5x1_f32x2 raft_convc2_f32x2 raft_fh1_f32x2 enc_3x3_256_384_f16 f16_3x3_256_512 dcn_offset_f16 fc1_f16 qkv_f16 fc2_f16 proj_f16 rfc_step_f16 rfc_off0_f16 rfc_bb2_f16 rfc_dcn_f16 featprop_bb2_f16 dec_3x3_128_128_f16 2>&1 | tee $O/convbench.log
cp $L /tmp/product.so; cp tools/variants/trace.so $L
timeout 60 tools/convbench raft_gru_1x5_f32x2 raft_convc2_f32x2 raft_fh1_f32x2 2>&1 | grep halo_trace | tee $O/halo_trace.log
cp /tmp/product.so $L
timeout 900 python -m pytest tests/test_conv.py tests/test_sample_kernels.py tests/test_raft.py tests/test_raft_kernels.py tests/test_rfc.py tests/test_generator.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench.err | grep "^{" | tail -1 > $O/bench.json; python -c "import json,sys;b=json.load(open('$O/bench.json'));print(b['value'], b['ms_per_step'], 'enqueue', b['host_enqueue_ms'], b['roofline']['frac'], b['roofline']['other'], b['parity']['psnr_db'], b['parity']['max_lsb'], b['parity']['flow_max_px'], b['node_call_frames_per_s'])" 2>&1 | tee $O/bench_summary.log
