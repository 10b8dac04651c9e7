#!/bin/bash
# Round 5: hierarchical kernel maps (SD3D_HIER_MAPS) vs the hash-probed maps, whole bench, alternating.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r05_maps_ab.txt"
: > "$OUT"
for v in 0 1 0 1; do
  echo "=== bench.py --steps 20 --warmup 5, SD3D_HIER_MAPS=$v" >> "$OUT"
  SD3D_HIER_MAPS=$v timeout 600 python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value', d['value'], 'single', d['single_scene']['latency_ms'], d['single_scene']['latency_ms_groups_of_5'], 'sustained', d['sustained']['scenes_per_s'], 'conv ms', r['ms_per_forward'], 'frac', r['frac'])
" >> "$OUT"
done
cat "$OUT"
