#!/bin/bash
# same-box A/B of two builds of the library (SD3D_LIB): ab_lib.sh <libA.so> <libB.so> [bench args]   (paths relative to the repo root)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in "$A" "$B"; do
    SD3D_LIB="$PWD/$L" python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-end-to-end --sustain-seconds 1.0 "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$L', 'value', d['value'], 'sustained', d['sustained']['scenes_per_s'], 'single ms', d['single_scene']['latency_ms'], 'conv ms', r['ms_per_forward'], 'frac', r['frac'])"
  done
done
