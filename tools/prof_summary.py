#!/usr/bin/env python3
"""Condense a rocprofv3 `*_results.db` (rocpd sqlite) into a small text summary for profiles/.

    python tools/prof_summary.py <results.db> <out.md> [--scenes N] [--delete]

Sections: per-kernel time (calls, total, average, share), the dominant kernel broken down by launch
shape, and - when the run collected PMC counters - per-kernel counter sums and per-dispatch averages.
"""
import argparse
import os
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("out")
    ap.add_argument("--scenes", type=int, default=0, help="forwards in the run (prints per-scene columns)")
    ap.add_argument("--delete", action="store_true", help="remove the .db afterwards (gpurun_out is size-capped)")
    ap.add_argument("--title", default="")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cur = db.cursor()
    lines = [f"# {args.title or os.path.basename(args.db)}", ""]
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    total_us = sum(r[2] for r in rows)
    lines += [f"total kernel time {total_us / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches"
              + (f" = {total_us / 1e3 / args.scenes:.3f} ms / forward ({args.scenes} forwards)" if args.scenes else ""), "",
              "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, tot, avg, pct in rows[:45]:
        lines.append(f"| `{name[:90]}` | {calls} | {tot / 1e3:.3f} | {avg:.1f} | {pct:.2f} |")
    n_stats = cur.execute("select count(*) from kernels where name like 'scene_stats_final%'").fetchone()[0]
    lines += ["", f"scenes in the run: {n_stats // 2} (scene_stats_final launches / 2: one for the scene range, one for the voxel keys, per scene - "
                  "whether a forward takes one scene or several)"]
    lines += ["", "## GEMM family by launch shape", "", "| variant | workgroups | launches | avg us | total ms | vgpr | lds |",
              "|---|---:|---:|---:|---:|---:|---:|"]
    q = ("select name, grid_x / workgroup_x, count(*), avg(duration), sum(duration), vgpr_count, lds_size, grid_y from kernels "
         "where name like '%gather_gemm%' or name like '%pair_gemm%' or name like '%pair_reduce%' "
         "group by name, grid_x, grid_y order by sum(duration) desc")
    for name, wgs, n, avg, tot, vg, lds, gy in cur.execute(q).fetchall()[:40]:
        short = name.split("(")[0].replace("void ", "")
        lines.append(f"| `{short}` | {wgs}x{gy} | {n} | {avg / 1e3:.1f} | {tot / 1e6:.3f} | {vg} | {lds} |")
    try:
        pm = cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        ctrs = sorted({r[1] for r in pm})
        table = {}
        for name, c, n, s in pm:
            table.setdefault(name, {})[c] = (n, s)
        dur = {r[0]: r[2] for r in rows}
        lines += ["", "## PMC counters (sum over all dispatches of the kernel)", "",
                  "| kernel | dispatches | " + " | ".join(ctrs) + " |", "|---|---:|" + "---:|" * len(ctrs)]
        for name in sorted(table, key=lambda k: -dur.get(k, 0))[:25]:
            n = max(v[0] for v in table[name].values())
            vals = [f"{table[name][c][1]:.4g}" if c in table[name] else "" for c in ctrs]
            lines.append(f"| `{name[:70]}` | {n} | " + " | ".join(vals) + " |")
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    db.close()
    if args.delete:
        os.remove(args.db)
    print(f"wrote {args.out}")


if __name__ == "__main__":
    main()
