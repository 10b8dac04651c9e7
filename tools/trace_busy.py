"""GPU busy time of a rocprofv3 kernel trace (rocpd sqlite): sum of kernel durations, union of their intervals, and the span from
the first dispatch to the last end - per window of the run when --split N is given (N equal slices of the dispatch sequence).
    python tools/trace_busy.py <results.db> [--skip-first FRACTION]"""
import argparse, sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--skip", type=float, default=0.3, help="fraction of the dispatches to drop from the front (warm-up)")
a = ap.parse_args()
db = sqlite3.connect(a.db)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
rows = []
for t in tabs:
    rows += db.execute(f"select start, end from {t}").fetchall()
rows.sort()
rows = rows[int(len(rows) * a.skip):]
total = sum(e - s for s, e in rows)
union, cur_s, cur_e = 0, None, None
gaps = []
for s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
            gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"dispatches {len(rows)}, sum of kernel durations {total / 1e6:.2f} ms, busy (union) {union / 1e6:.2f} ms, span {span / 1e6:.2f} ms "
      f"-> GPU busy {100 * union / span:.1f} % of the span")
gaps.sort(reverse=True)
big = [g for g in gaps if g > 20000]
print(f"idle gaps: {len(gaps)} in total, {sum(gaps) / 1e6:.2f} ms; gaps > 20 us: {len(big)} = {sum(big) / 1e6:.2f} ms; largest {[round(g / 1e3) for g in gaps[:8]]} us")
