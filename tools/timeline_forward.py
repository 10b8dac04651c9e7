#!/usr/bin/env python3
"""Kernel timeline of the LAST forward in a rocprofv3 --kernel-trace database of tools/single_forward.py:
   python tools/timeline_forward.py <results.db> [out.md]
Per forward: span (first kernel start -> last kernel end), busy time of the device (union of kernel intervals), per-queue busy time,
and the idle gaps of the union > 15 us with the kernels around them."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
print("kernels columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f"select name, start, end, {qcol or '0'} from kernels order by start"
rows = cur.execute(sel).fetchall()
# forwards: split at gaps > 2 ms... the single-scene loop has no such gaps; split by the scene_stats kernel (first launch of a forward)
starts = [i for i, r in enumerate(rows) if r[0].startswith("scene_stats_kernel") or r[0].startswith("scene_stats(")]
if not starts:
    starts = [i for i, r in enumerate(rows) if "scene_stats" in r[0]]
# a forward launches scene_stats twice in a row at most; keep the first of each cluster (> 1 ms apart)
fw = []
for i in starts:
    if not fw or rows[i][1] - rows[fw[-1]][1] > 3_000_000:
        fw.append(i)
out = []
for k in range(max(0, len(fw) - 4), len(fw) - 1):
    seg = rows[fw[k]:fw[k + 1]]
    t0 = seg[0][1]
    span = (seg[-1][2] - t0) / 1e3
    period = (rows[fw[k + 1]][1] - t0) / 1e3
    # union of intervals
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for name, s, e, q in seg:
        if cur_e is None:
            cur_s, cur_e, last = s, e, name
        elif s <= cur_e:
            if e > cur_e:
                cur_e, last = e, name
        else:
            busy += cur_e - cur_s
            gaps.append(((s - cur_e) / 1e3, (cur_e - t0) / 1e3, last, name))
            cur_s, cur_e, last = s, e, name
    busy += cur_e - cur_s
    per_q = {}
    for name, s, e, q in seg:
        per_q[q] = per_q.get(q, 0) + (e - s)
    out.append(f"forward {k}: {len(seg)} kernels, period {period:.0f} us (start to next forward's start), span {span:.0f} us, device busy (union) {busy / 1e3:.0f} us, "
               f"sum of kernel times {sum(e - s for _, s, e, _ in seg) / 1e3:.0f} us; per queue: " + ", ".join(f"{q}: {v / 1e3:.0f} us" for q, v in sorted(per_q.items())))
    big = sorted(gaps, reverse=True)[:14]
    tot_gap = sum(g[0] for g in gaps)
    out.append(f"   idle (no kernel on any queue) {tot_gap:.0f} us in {len(gaps)} gaps; gaps > 4 us: {sum(1 for g in gaps if g[0] > 4)} = {sum(g[0] for g in gaps if g[0] > 4):.0f} us; the largest:")
    for g, at, a, b in sorted(big, key=lambda x: x[1]):
        out.append(f"     {g:7.1f} us at +{at:8.0f} us  after `{a[:50]}` before `{b[:50]}`")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
if len(sys.argv) > 3:                                          # list the kernels of the last complete forward from the first kernel whose name contains argv[3]
    seg = rows[fw[-2]:fw[-1]]
    t0 = seg[0][1]
    start = next((i for i, r in enumerate(seg) if sys.argv[3] in r[0]), 0)
    lines = [f"kernels of the last complete forward from `{sys.argv[3]}` on ({len(seg) - start} launches):"]
    for name, s_, e_, q in seg[start:]:
        lines.append(f"  +{(s_ - t0) / 1e3:8.1f} us  {(e_ - s_) / 1e3:6.1f} us  q{q}  {name[:70]}")
    print("\n".join(lines))
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write("\n".join(lines) + "\n")
