#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2 3; do
  for s in 3 4 5; do
    r=$(python bench.py --no-cpu-baseline --steps 60 --warmup 8 --streams $s 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")
    echo "streams $s : $r"
  done
done
