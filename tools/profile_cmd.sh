#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool, condensed.  usage: profile_cmd.sh <name> <script.py> [args...]
NAME=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
SCRIPT=$1; shift
rocprofv3 --kernel-trace --stats -d /tmp/prof_$NAME -o r -- python3 "$ROOT/$SCRIPT" "$@" > "$ROOT/gpurun_out/$NAME.out" 2> /tmp/prof_$NAME.err
DB=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python3 "$ROOT/tools/prof_summary.py" "$DB" "$ROOT/gpurun_out/$NAME.md" --delete --title "$NAME"
