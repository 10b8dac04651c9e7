#!/bin/bash
# PMC pass of bench.py (single stream), condensed.   usage: PMC="FETCH_SIZE" pmc_run.sh <name> [bench args]
NAME=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
rocprofv3 --kernel-trace --pmc $PMC -d /tmp/prof_$NAME -o r -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$ROOT/gpurun_out/$NAME.bench.json" 2> /tmp/prof_$NAME.err
DB=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python3 "$ROOT/tools/prof_summary.py" "$DB" "$ROOT/gpurun_out/$NAME.md" --delete --title "$NAME: rocprofv3 --kernel-trace --pmc $PMC -- python3 bench.py --no-cpu-baseline $*" || tail -5 /tmp/prof_$NAME.err
