#!/bin/bash
# SQ / GRBM counter passes over the attention kernels of the tree (tools/attn_pmc.py, one query count per process so that a kernel
# name maps to one launch shape).  The program goes directly after `--`.  -> gpurun_out/pmc_attn_q<Q>_p<pass>.json
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
for q in ${1:-200 3000}; do
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    rm -rf /tmp/pa
    rocprofv3 --kernel-trace --pmc $P -d /tmp/pa -o r -- python3 "$ROOT/tools/attn_pmc.py" $q > "$ROOT/gpurun_out/pmc_attn_q${q}_p${i}.out" 2> /tmp/pa.err
    DB=$(find /tmp/pa -name "*.db" | head -1)
    if [ -n "$DB" ]; then python3 "$ROOT/tools/pmc_collect.py" "$DB" "$ROOT/gpurun_out/pmc_attn_q${q}_p${i}.json" attention; else tail -3 /tmp/pa.err; fi
  done
done
