#!/bin/bash
# Is the clip power-limited?  Sample board power and clocks (rocm-smi) once a second while bench.py times 20 steps.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|mclk" | head -8
(timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/power_bench.json 2>/dev/null) &
BP=$!
sleep 25   # import + weights + warm-up
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Graphics Package Power\|Socket\|sclk" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
python -c "import json;b=json.loads([l for l in open('gpurun_out/power_bench.json').read().splitlines() if l.startswith('{')][-1]);print(b['value'], b['ms_per_step'])"
