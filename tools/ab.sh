#!/bin/bash
# same-box A/B of two environment settings: ab.sh "ENV_A=.." "ENV_B=.." [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A="$1"; B="$2"; shift 2
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    r=$(env $v python bench.py --no-cpu-baseline --steps 48 --warmup 8 "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j.get('single_scene', {}).get('latency_ms'))")
    echo "$v : $r"
  done
done
