#!/bin/bash
# Host-bound or GPU-bound?  Two independent bench processes on the same GPU vs one.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py --no-cpu-baseline --steps 40 --warmup 6 > /tmp/one.json 2>/dev/null
echo "one process:"; cat /tmp/one.json | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 40 --warmup 6 > /tmp/a.json 2>/dev/null &
P1=$!
python bench.py --no-cpu-baseline --steps 40 --warmup 6 > /tmp/b.json 2>/dev/null &
P2=$!
wait $P1 $P2
echo "two processes (each):"
for f in /tmp/a.json /tmp/b.json; do cat $f | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
