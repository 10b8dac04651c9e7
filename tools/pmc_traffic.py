#!/usr/bin/env python3
"""Condense the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs of bench.py --streams 1, summarised by
tools/prof_summary.py) into profiles/<round>_pmc_traffic.json: HBM-side bytes per forward of the dominant kernel (pair_gemm_* +
pair_reduce_kernel) and of the whole GEMM family.   usage: pmc_traffic.py <fetch.md> <write.md> <out.json> [<bench.json of one of the passes>]
The bench line of the profiled run (tools/pmc_run.sh keeps it next to the summary) supplies the WORKLOAD KEY - scene shape, layout,
scenes per forward - that bench.py compares with its own run before it reports these bytes.

Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) x 2 on gfx950 (128-B requests tallied at 64 B; calibrated for
16 B/lane streaming reads, which is what the gathers and the pass-2 reads are), WRITE_SIZE (KB) as reported."""
import json
import re
import sys


def table(path):
    rows, forwards = {}, 0
    in_pmc = False
    for line in open(path):
        if line.startswith("## PMC counters"):
            in_pmc = True
            continue
        m = re.match(r"\| `(.+?)` \| (\d+) \| (.+) \|", line)
        if in_pmc and m:
            vals = [v.strip() for v in m.group(3).split("|")]
            rows[m.group(1)] = (int(m.group(2)), float(vals[0]) if vals[0] else 0.0)
        ms = re.match(r"scenes in the run: (\d+)", line)
        if ms:
            forwards = int(ms.group(1))              # SCENES of the run (tools/prof_summary.py), one scene or several per forward
    return rows, forwards


def main():
    fetch, nf = table(sys.argv[1])
    write, nw = table(sys.argv[2])
    assert nf and nw, (nf, nw)                      # scenes of the two passes (the time-based pre-roll may give them different counts)
    conv = lambda n: n.startswith("pair_gemm") or n.startswith("pair_reduce") or n.startswith("pair_center")  # noqa: E731
    fam = lambda n: conv(n) or "gather_gemm" in n  # noqa: E731
    import os
    tracked = lambda p: "profiles/" + os.path.basename(p)     # the summaries are committed under profiles/ (gpurun_out/ is scratch)  # noqa: E731
    out = {"source": f"{tracked(sys.argv[1])} + {tracked(sys.argv[2])} (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes, bench.py --streams 1)",
           "forwards": nf, "forwards_write_pass": nw, "forwards_note": "scenes of the profiled runs (bytes_per_forward = bytes per scene)",
           "fetch_correction": "x2 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half the bytes of 16 B/lane reads)",
           "write_correction": "none (uncalibrated)"}
    for key, sel in (("conv", conv), ("family", fam)):
        f_kb = sum(v[1] for n, v in fetch.items() if sel(n))
        w_kb = sum(v[1] for n, v in write.items() if sel(n))
        out[f"{key}_fetch_kb_sum"], out[f"{key}_write_kb_sum"] = f_kb, w_kb
        out[f"{key}_bytes_per_forward"] = int((2.0 * f_kb / nf + w_kb / nw) * 1024)
    out["bytes_per_forward"] = out["family_bytes_per_forward"]
    out["pass1_launches_per_scene"] = round(sum(v[0] for n, v in fetch.items() if n.startswith("pair_gemm")) / nf, 2)
    if len(sys.argv) > 4:
        for line in open(sys.argv[4]):
            if line.startswith("{"):
                cfg = json.loads(line)["config"]
                out["workload"] = {"points": cfg["points"], "superpoints": cfg["superpoints"], "queries_2d": cfg["queries_2d"],
                                   "scene_layout": cfg["scene_layout"],
                                   "forward_sizes": sorted({n for st in cfg["forward_sizes"] for n in st})}   # sizes of the profiled forwards
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
