#!/bin/bash
# r06 GPU call 2: the patch-gathering PP_F32X2 convolution (conv_patch.hip), RAFT without its cat / stack copies, reference-frame
# tokens once per clip, feature-propagation inputs gathered into the graph's statics: kernel + stage + e2e tests, then the
# bench line with and without the patch kernel.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call2; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 1200 python -m pytest tests/test_conv.py tests/test_raft.py tests/test_raft_kernels.py tests/test_generator.py tests/test_e2e.py tests/test_distributed.py tests/test_nodes.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.log
timeout 600 python -m pytest tests/test_baseline_configs.py -x -q -m gpu -k "cfg2_24f or cfg1 or cfg2_80f_node or cfg3_12f" -s 2>&1 | grep -v "^$" | tail -30 | cut -c1-400 | tee $O/pytest_cfg.log
PP_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_timing.json 2> $O/bench_timing.err
grep "stage ms" $O/bench_timing.json $O/bench_timing.err | tail -2 | cut -c1-600
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
PP_CONV_PATCH=0 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_nopatch.json 2> $O/bench_nopatch.err
for f in bench bench_nopatch; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d.get('node_call_frames_per_s'), d.get('host_enqueue_ms'), d.get('parity'))
PY
done
