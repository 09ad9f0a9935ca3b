#!/bin/bash
# r06 GPU call 10: RAFT's two directions as two hipGraphs on two streams (PP_RAFT_LANES): RAFT tests, bit-identity, A/B bench.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call10; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 600 python -m pytest tests/test_raft.py tests/test_e2e.py tests/test_graphs.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4
import os, sys, torch
sys.path.insert(0, '.')
from comfyui_propainter_nodes_amd import raft, weights, synth
dev = torch.device('cuda:0')
image, _ = synth.synthetic_clip(12, 360, 640)
frames = (image * 2 - 1).to(dev).contiguous()
R = raft.RaftFlow(weights.synth_state_dicts(0)['raft'], dev)
os.environ['PP_GRAPHS_RAFT'] = '0'
a = R.bidirectional(frames, 6).clone()
os.environ['PP_GRAPHS_RAFT'] = '1'
for lanes in ('1', '2', '2'):
    os.environ['PP_ENC_LANES'] = lanes
    b = R.bidirectional(frames, 6)
    print('lanes', lanes, 'graph == eager:', bool(torch.equal(a, b)), float((a - b).abs().max()))
PY
for v in 2 1 2 1; do
  PP_FEATPROP_LANES=$v PP_ENC_LANES=$v timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("FEATPROP+ENC LANES", sys.argv[2], d['value'], d['ms_per_step'], d.get('host_enqueue_ms'), d['parity']['psnr_db'], d['parity']['max_lsb'], d['parity']['flow_max_px'])
PY
done
