#!/usr/bin/env python3
"""rocprofv3 results.db (rocpd sqlite) -> JSON: per kernel name {dispatches, avg_us, counters: {name: sum over dispatches}}.
    python tools/pmc_collect.py <results.db> <out.json> [name-substring ...]"""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
subs = sys.argv[3:]
out = {}
for name, n, avg in cur.execute("select name, count(*), avg(duration) from kernels group by name").fetchall():
    if subs and not any(s in name for s in subs):
        continue
    out[name] = {"dispatches": n, "avg_us": avg / 1e3, "counters": {}}
try:
    for name, c, n, s in cur.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall():
        if name in out:
            out[name]["counters"][c] = s
            out[name]["counter_rows"] = n
except sqlite3.Error as e:
    out["_error"] = str(e)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2], len(out), "kernels")
