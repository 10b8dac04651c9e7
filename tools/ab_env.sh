#!/bin/bash
# same-box A/B of environment settings over bench.py: ab_env.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   (each setting run twice, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS=$1; shift
for rep in 1 2; do
  for E in "$@"; do
    env $E python bench.py --no-cpu-baseline --no-end-to-end --sustain-seconds 0.7 $ARGS 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$E |', 'value', d['value'], 'sustained', d['sustained']['scenes_per_s'], 'single ms', d['single_scene']['latency_ms'], 'conv ms', r['ms_per_forward'], 'frac', r['frac'])"
  done
done
