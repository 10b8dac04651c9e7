#!/bin/bash
# rocprofv3 PMC pass (counters in $PMC) of a python tool, condensed.  usage: PMC="A B C" pmc_cmd.sh <name> <script.py> [args...]
NAME=$1; shift
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
SCRIPT=$1; shift
rocprofv3 --kernel-trace --pmc $PMC -d /tmp/prof_$NAME -o r -- python3 "$ROOT/$SCRIPT" "$@" > "$ROOT/gpurun_out/$NAME.out" 2> /tmp/prof_$NAME.err
DB=$(find /tmp/prof_$NAME -name "*.db" | head -1)
python3 "$ROOT/tools/prof_summary.py" "$DB" "$ROOT/gpurun_out/$NAME.md" --delete --title "$NAME ($PMC)" || tail -5 /tmp/prof_$NAME.err
