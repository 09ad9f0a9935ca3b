#!/bin/bash
# r06 GPU call 9: pairs of transformer windows on two streams (PP_WINDOW_PAIRS): bit-identity against the serial form, A/B bench.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_call9; mkdir -p $O
export PP_ALLOW_SYNTHETIC_WEIGHTS=1
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -5
import os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from comfyui_propainter_nodes_amd import pipeline, weights, synth, image_utils
dev = torch.device('cuda:0')
T, H, W = 40, 360, 640
image, mask = synth.synthetic_clip(T, H, W)
fr, fm, md = image_utils.prepare_frames_and_masks(image_utils.image_to_uint8_frames(image), mask, image_utils.ImageConfig(W, H, 5, 8, (W, H), T))
models = pipeline.models_from_state_dicts(weights.synth_state_dicts(0), dev)
cfg = pipeline.ProPainterConfig(10, 10, 80, 5, 'enable', T, dev, (W, H))
os.environ['PP_WINDOW_LANES'] = '1'
a = pipeline.run_inpainting(models, fr, fm, md, cfg)
os.environ['PP_WINDOW_LANES'] = '3'
for i in range(3):
    b = pipeline.run_inpainting(models, fr, fm, md, cfg)
    print('pairs == serial:', bool(torch.equal(a, b)))
PY
for v in 2 3 4 1 2 3; do
  PP_WINDOW_LANES=$v timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("LANES", sys.argv[2], d['value'], d['ms_per_step'], d.get('host_enqueue_ms'), d['parity']['psnr_db'], d['parity']['max_lsb'])
PY
done
