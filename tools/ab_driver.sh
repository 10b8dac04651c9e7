#!/bin/bash
# the driver's command line (python3 bench.py --gpus 1 --steps 20 --warmup 5) for several (streams, batch) settings, same box
cd "$(dirname "$0")/.."
for sb in $1; do
  s=${sb%,*}; b=${sb#*,}
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --streams $s --batch $b ${@:2} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('driver line, streams $s batch $b:', d['value'], 'scenes/s | single', d['single_scene']['latency_ms'], 'ms | conv frac', r['frac'], 'ms/fwd', r['ms_per_forward'])
"
done
