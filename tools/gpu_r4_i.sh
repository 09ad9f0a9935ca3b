#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ONLY="off6,via the halo,off2  3x3,dcn" REPS=600 timeout 300 python tools/diag_kernels_under_load.py 2>&1 | grep "under load" | cut -c1-150
timeout 300 python tools/diag_rfc_under_load.py 2>&1 | grep "traced\|under load" | cut -c1-200
timeout 300 python tools/diag_overlap_identity.py 640x360x200 2>&1 | grep "rep\|completion" | cut -c1-260
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);r=b['roofline'];print(b['value'], b['ms_per_step'], r['achieved'], r['frac'], r['other']['f16'], r['attention']['achieved'], r['attention']['avg_launch_us'], b['parity']['psnr_db'], b['parity']['max_lsb'], b['parity']['flow_max_px'])"
