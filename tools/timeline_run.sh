#!/bin/bash
NAME=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -o r -- python3 "$ROOT/bench.py" --no-cpu-baseline "$@" > /dev/null 2> /tmp/prof_$NAME.err
DB=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python3 "$ROOT/tools/timeline.py" "$DB"
