#!/bin/bash
# timeline_cmd.sh <name> <script.py> [args]: rocprofv3 kernel trace + tools/timeline.py
NAME=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
SCRIPT=$1; shift
rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -o r -- python3 "$ROOT/$SCRIPT" "$@" 2> /tmp/prof_$NAME.err | tail -1
DB=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python3 "$ROOT/tools/timeline.py" "$DB" | tail -1
