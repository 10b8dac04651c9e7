#!/bin/bash
# the driver's command line, N times on one box: value / single_scene / conv fraction of every run (the 20-step line is noisy)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out
N=${1:-6}
: > gpurun_out/r05_driverline_repeats.txt
for i in $(seq 1 $N); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('run $i: value', d['value'], 'ms_per_step', d['ms_per_step'], 'single_scene', (d.get('single_scene') or {}).get('latency_ms'), 'conv frac', r['frac'], 'conv ms', r['ms_per_forward'])
" | tee -a gpurun_out/r05_driverline_repeats.txt
done
