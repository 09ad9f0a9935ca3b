#!/bin/bash
# A/B of a library variant (tools/build_variant.sh) against the product on tools/bench_conv.py's RAFT shapes + parity tests + bench
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=${1:-xfetch}
L=comfyui_propainter_nodes_amd/libpropainter_mi355.so
cp $L /tmp/product.so
echo "== product"; timeout 120 python tools/bench_conv.py 2>/dev/null | grep "f32x2"
timeout 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;b=json.loads(sys.stdin.read());print(b['value'], b['ms_per_step'], b['roofline']['achieved'], b['roofline']['frac'], b['parity']['psnr_db'], b['parity']['flow_max_px'])"
cp tools/variants/$V.so $L
echo "== variant $V"; timeout 120 python tools/bench_conv.py 2>/dev/null | grep "f32x2"
timeout 300 python -m pytest tests/test_conv.py -q -m gpu -k "matches_torch and halo" 2>&1 | tail -2
timeout 200 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;b=json.loads(sys.stdin.read());print(b['value'], b['ms_per_step'], b['roofline']['achieved'], b['roofline']['frac'], b['parity']['psnr_db'], b['parity']['flow_max_px'])"
cp /tmp/product.so $L
