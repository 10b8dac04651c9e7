#!/bin/bash
# MFMA-pipe utilisation of pass 1 on one layer (PAIR_CASES index of tools/pair_quick.py) under ablation switches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PAIR_CASES=${PAIR_CASES:-6}
for d in "$@"; do
  SD3D_PAIR_DBG=$d PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" bash tools/pmc_cmd.sh pp$d tools/pair_quick.py > /dev/null 2>&1
  echo "DBG=$d"; grep "pair_gemm" gpurun_out/pp$d.md | head -4
done
