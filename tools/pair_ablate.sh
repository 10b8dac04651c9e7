#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for d in "$@"; do
  SD3D_PAIR_DBG=$d bash tools/profile_cmd.sh pa$d tools/pair_quick.py > /dev/null 2>&1
  echo "DBG=$d"; grep "pair_gemm_" gpurun_out/pa$d.md | grep "x" | sort | awk -F"|" '{printf "%s %s %s; ", $2, $3, $5}'; echo
done
