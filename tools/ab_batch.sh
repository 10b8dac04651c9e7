#!/bin/bash
# same-box A/B of (streams, batch) settings of bench.py; usage: tools/ab_batch.sh "4,1 2,4 1,4" [extra bench args]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for sb in $1; do
  s=${sb%,*}; b=${sb#*,}
  python bench.py --no-cpu-baseline --steps 96 --warmup 3 --streams $s --batch $b ${@:2} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('streams $s batch $b:', d['value'], 'scenes/s | single', d['single_scene']['latency_ms'], 'ms | conv frac', r['frac'], 'ms/fwd', r['ms_per_forward'])
"
done
