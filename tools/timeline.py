#!/usr/bin/env python3
"""GPU timeline statistics from a rocprofv3 results.db: how much of the wall time has at least one
kernel running, how much has a 'big' (pair_gemm / pair_reduce) kernel running, idle gaps."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]

rows = cur.execute("select name, start, end from kernels order by start").fetchall()
print("dispatches", len(rows))
# skip the first 40 % (warm-up) and analyse the rest
t0 = rows[int(len(rows) * 0.4)][1]
rows = [r for r in rows if r[1] >= t0]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
span = max(r[2] for r in rows) - rows[0][1]
allk = union([(r[1], r[2]) for r in rows])
big = [(r[1], r[2]) for r in rows if "pair_gemm" in r[0] or "pair_reduce" in r[0]]
bigu = union(big)
print(f"span {span/1e6:.2f} ms; any kernel running {100*allk/span:.1f} %; big kernel running {100*bigu/span:.1f} %; sum of big durations {sum(e-s for s,e in big)/1e6:.2f} ms; sum of all durations {sum(r[2]-r[1] for r in rows)/1e6:.2f} ms")
