#!/bin/bash
# SQ / TCC / GRBM counter passes over the pair-major convolution of single U-Net layers (tools/pair_quick.py, one layer shape per
# process so that a kernel name maps to one launch shape).  The program goes directly after `--`.
#   usage: tools/pmc_pair.sh "<case indices of pair_quick.py>"      -> gpurun_out/pmc_pair_c<case>_p<pass>.json
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P3="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"
P4="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
for c in $1; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    rm -rf /tmp/pp
    PAIR_CASES=$c rocprofv3 --kernel-trace --pmc $P -d /tmp/pp -o r -- python3 "$ROOT/tools/pair_quick.py" > "$ROOT/gpurun_out/pmc_pair_c${c}_p${i}.out" 2> /tmp/pp.err
    DB=$(find /tmp/pp -name "*.db" | head -1)
    if [ -n "$DB" ]; then python3 "$ROOT/tools/pmc_collect.py" "$DB" "$ROOT/gpurun_out/pmc_pair_c${c}_p${i}.json" pair_gemm pair_reduce pair_center; else tail -3 /tmp/pp.err; fi
  done
done
