#!/bin/bash
# Round 5: lock-step pass 1, first form vs second form (SD3D_PAIR_V2) and time-sliced priority (SD3D_PAIR_FAIR), layer by layer
# (tools/pair_quick.py, one process per setting: the library reads its switches once) and in the whole bench.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r05_pair_ab.txt"
: > "$OUT"
run() {   # label, env...
  label=$1; shift
  echo "=== $label" >> "$OUT"
  env "$@" PAIR_CHAINED=product timeout 300 python3 "$ROOT/tools/pair_quick.py" >> "$OUT" 2>&1
}
run "single-scene choice, V2=0" SD3D_PAIR_V2=0
run "single-scene choice, V2=1" SD3D_PAIR_V2=1
run "single-scene choice, V2=1 FAIR=13" SD3D_PAIR_V2=1 SD3D_PAIR_FAIR=13
run "crowded choice (all lock-step), V2=0" SD3D_PAIR_V2=0 SD3D_PAIR_CROWD=1
run "crowded choice (all lock-step), V2=1" SD3D_PAIR_V2=1 SD3D_PAIR_CROWD=1
run "crowded choice (all lock-step), V2=1 FAIR=13" SD3D_PAIR_V2=1 SD3D_PAIR_CROWD=1 SD3D_PAIR_FAIR=13
run "crowded choice (all lock-step), V2=1 FAIR=15" SD3D_PAIR_V2=1 SD3D_PAIR_CROWD=1 SD3D_PAIR_FAIR=15
run "crowded choice (all lock-step), V2=0 FAIR=13" SD3D_PAIR_V2=0 SD3D_PAIR_CROWD=1 SD3D_PAIR_FAIR=13
for v in 0 1 0 1; do
  echo "=== bench.py --steps 20 --warmup 5, V2=$v" >> "$OUT"
  SD3D_PAIR_V2=$v timeout 600 python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('value', d['value'], 'single', d['single_scene']['latency_ms'], 'sustained', d['sustained']['scenes_per_s'], 'conv ms', r['ms_per_forward'], 'frac', r['frac'])
" >> "$OUT"
done
tail -100 "$OUT"
