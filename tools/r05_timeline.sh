#!/bin/bash
# kernel timeline of one single-scene forward (all launches, start offsets, queue)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace -d /tmp/prof_tl -o r -- python3 "$ROOT/tools/single_forward.py" 6 2> /tmp/prof_tl.err | tail -1
DB=$(find /tmp/prof_tl -name "*.db" | head -1)
python3 "$ROOT/tools/timeline_forward.py" "$DB" "$ROOT/gpurun_out/r05_timeline_single.md" scene_stats > /dev/null
head -30 "$ROOT/gpurun_out/r05_timeline_single.md"
