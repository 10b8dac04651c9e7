#!/bin/bash
# wave-lifetime counters of pass 1 with and without the shared tail (SD3D_PAIR_POOL), layers 1 / 4 / 7 of tools/pair_quick.py
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES GRBM_GUI_ACTIVE"
for pool in 1 0; do
  for c in 1 4 7; do
    rm -rf /tmp/pp
    export SD3D_PAIR_POOL=$pool PAIR_CHAINED=product PAIR_CASES=$c
    rocprofv3 --kernel-trace --pmc $P1 -d /tmp/pp -o r -- python3 "$ROOT/tools/pair_quick.py" > /dev/null 2> /tmp/pp.err
    DB=$(find /tmp/pp -name "*.db" | head -1)
    if [ -n "$DB" ]; then python3 "$ROOT/tools/pmc_collect.py" "$DB" "$ROOT/gpurun_out/pmc_pool${pool}_c${c}.json" pair_gemm; else tail -3 /tmp/pp.err; fi
  done
done
python3 - <<'PY'
import json, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
names = {1: "level 0 96->96", 4: "level 1 96->96", 7: "level 2 192->128"}
out = ["| layer | shared tail | avg us | MFMA pipe busy % of kernel | mean wave lifetime % of kernel | busy % of lifetime |", "|---|---|---:|---:|---:|---:|"]
for c in (1, 4, 7):
    for pool in (0, 1):
        d = json.load(open(f"{root}/pmc_pool{pool}_c{c}.json"))
        for k, v in d.items():
            if k.startswith("_") or "pair_gemm" not in k: continue
            n = v["dispatches"]; e = {cn: cv / n for cn, cv in v["counters"].items()}
            cyc = e["GRBM_GUI_ACTIVE"] / 8; busy = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024; life = 4 * e["SQ_WAVE_CYCLES"] / e["SQ_WAVES"]
            out.append(f"| {names[c]} | {'on' if pool else 'off'} | {v['avg_us']:.1f} | {100 * busy / cyc:.1f} | {100 * life / cyc:.1f} | {100 * busy / life:.1f} |")
open(f"{root}/r05_pmc_pool.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
