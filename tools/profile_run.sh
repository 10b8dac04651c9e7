#!/bin/bash
# rocprofv3 kernel trace of bench.py, condensed to gpurun_out/<name>.md.   usage: profile_run.sh <name> [bench args...]
NAME=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -o r -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > "$ROOT/gpurun_out/$NAME.bench.json" 2> /tmp/prof_$NAME.err
DB=$(find /tmp/prof_$NAME -name "*.db" | head -1)
STEPS=13
python3 "$ROOT/tools/prof_summary.py" "$DB" "$ROOT/gpurun_out/$NAME.md" --delete --title "$NAME: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline $*"
