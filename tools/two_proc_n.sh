#!/bin/bash
# two_proc_n.sh <procs> <streams>: P independent bench processes on one GPU, S streams each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=$1; S=$2
pids=()
for i in $(seq 1 $P); do
  python bench.py --no-cpu-baseline --steps 60 --warmup 6 --streams $S > /tmp/tp_$i.json 2>/dev/null &
  pids+=($!)
done
wait "${pids[@]}"
python - <<PY
import json
tot=0
for i in range(1,$P+1):
    j=json.loads(open(f"/tmp/tp_{i}.json").read().strip().splitlines()[-1]); tot+=j["value"]
print("procs $P x streams $S : total %.1f scenes/s" % tot)
PY
