#!/bin/bash
# same-box A/B of the fused (row-chain) decoder against the op-by-op decoder: bench.py at 200 queries and one query per superpoint
# usage: fused_ab.sh [extra bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for q in 200 -1; do
  for f in 1 0; do
    SD3D_FUSED_DECODER=$f python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-end-to-end --query-num $q --sustain-seconds 1.0 "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('query_num', $q, 'fused', $f, 'value', d['value'], 'sustained', d['sustained']['scenes_per_s'], 'single_scene ms', d['single_scene']['latency_ms'])"
  done
done
