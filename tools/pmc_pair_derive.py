#!/usr/bin/env python3
"""Derived table from the counter passes of tools/pmc_pair.sh (gpurun_out/pmc_pair_c<case>_p<pass>.json):
matrix-pipe busy %, wave lifetime, wait shares, L2 hit rate, effective clock, per kernel and layer shape.
    python tools/pmc_pair_derive.py gpurun_out > profiles/r03_pmc_pair_gemm_table.md"""
import glob, json, os, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
CASES = {0: "stem k5 288->32 (level 0, P=990 k)", 1: "level 0 96->96 (P=424 k)", 4: "level 1 96->96 (P=777 k)",
         7: "level 2 192->128 (P=681 k)", 9: "level 3 256->256 (P=228 k)", 11: "level 4 256->256 (P=57 k)"}
FLOPS = {0: 2 * 990230 * 288 * 32, 1: 2 * 423582 * 96 * 96, 4: 2 * 777134 * 96 * 96, 7: 2 * 680872 * 192 * 128,
         9: 2 * 227930 * 256 * 256, 11: 2 * 56881 * 256 * 256}
N_SIMD, N_XCD = 1024, 8
rows = []
for c, label in CASES.items():
    d = {}
    for p in (1, 2, 3, 4):
        f = os.path.join(root, f"pmc_pair_c{c}_p{p}.json")
        if not os.path.exists(f):
            continue
        for k, v in json.load(open(f)).items():
            if k.startswith("_") or "pair_gemm" not in k:
                continue
            e = d.setdefault(k.split("(")[0], {"us": [], "n": v["dispatches"]})
            e["us"].append(v["avg_us"])
            for cn, cv in v["counters"].items():
                e[cn] = cv / v["dispatches"]
    for kern, e in d.items():
        us = sum(e["us"]) / len(e["us"])
        cyc = e["GRBM_GUI_ACTIVE"] / N_XCD                      # one count per XCD
        mhz = cyc / us
        busy = e["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD            # cycles per SIMD (64 per v_mfma_f32_32x32x2_f32)
        waves = e["SQ_WAVES"]
        life = 4 * e["SQ_WAVE_CYCLES"] / waves                   # quad-cycles -> cycles
        wait_any = 4 * e["SQ_WAIT_ANY"] / waves
        wait_inst = 4 * e["SQ_WAIT_INST_ANY"] / waves
        active = 4 * e["SQ_ACTIVE_INST_ANY"] / waves
        wps = waves / N_SIMD
        rows.append((label, kern, int(waves), us, mhz, FLOPS[c] / us / 1e6, 100 * busy / cyc, 100 * life / cyc,
                     100 * busy / life * (1 if wps <= 1 else 1), wps, 100 * wait_any / life, 100 * wait_inst / life, 100 * active / life,
                     100 * e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4 * e["SQ_WAIT_INST_LDS"] / waves / life * 100,
                     e["SQ_LDS_BANK_CONFLICT"], e["SQ_INSTS_MFMA"] * 4096 / FLOPS[c]))
print("| layer | pass-1 kernel | waves | avg us | clock MHz | TFLOP/s | MFMA pipe busy % of kernel | mean wave lifetime % of kernel | waves per SIMD |"
      " s_waitcnt / barrier % of lifetime | issue stall % of lifetime | issuing % of lifetime | LDS issue stall % | L2 hit % | LDS bank conflicts | MFMA flops / real flops |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for r in rows:
    print(f"| {r[0]} | `{r[1]}` | {r[2]} | {r[3]:.1f} | {r[4]:.0f} | {r[5]:.1f} | {r[6]:.1f} | {r[7]:.1f} | {r[9]:.0f} | {r[10]:.1f} | {r[11]:.1f} | {r[12]:.1f} | {r[14]:.2f} | {r[13]:.1f} | {r[15]:.0f} | {r[16]:.3f} |")
