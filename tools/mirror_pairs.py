"""How many rulebook entries of a stride-1 3^3 table come in MIRROR pairs (row r has a neighbour at +d AND at -d)?  If a mirror
pair's two products shared one partial-product row, the partial traffic of the pair-major convolution would drop by that share.
CPU only (numpy), benchmark and scan layouts.   usage: python tools/mirror_pairs.py [points]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from segdino3d_amd.synth import make_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
for layout in ("benchmark", "scan"):
    pts, _ = make_scene(0, n, 3000, 300, layout=layout)
    c0 = np.floor(pts[:, :3].numpy() * np.float32(50.0)).astype(np.int64)
    c0 -= c0.min(0) - 40
    print(f"layout {layout}: {n} points")
    tot_p = tot_new = 0
    for lvl in range(5):
        c = np.unique(c0 >> lvl, axis=0)
        key = lambda a: (a[:, 0] << 40) | (a[:, 1] << 20) | a[:, 2]  # noqa: E731
        keys = np.sort(key(c))
        has = lambda a: keys[np.clip(np.searchsorted(keys, key(a)), 0, len(keys) - 1)] == key(a)  # noqa: E731
        offs = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
        P = both = 0
        for i, o in enumerate(offs):
            h = has(c + np.array(o))
            P += int(h.sum())
            if i < 13:
                both += int((h & has(c - np.array(o))).sum())
        V = len(c)
        print(f"  level {lvl}: V = {V}, pairs / row {P / V:.2f}, mirror pairs with BOTH neighbours: {both} = {2 * both / max(1, P - V):.1%} of the non-centre "
              f"entries; partial rows {P} -> {P - both} ({1 - (P - both) / P:.1%} fewer)")


# ---- what larger groups would buy: partial rows per table when the 13 mirror lines are merged G at a time (G = 1: the chained lists as built)
print()
for layout in ("benchmark",):
    pts, _ = make_scene(0, n, 3000, 300, layout=layout)
    c0 = np.floor(pts[:, :3].numpy() * np.float32(50.0)).astype(np.int64)
    c0 -= c0.min(0) - 40
    for lvl in range(3):
        c = np.unique(c0 >> lvl, axis=0)
        key = lambda a: (a[:, 0] << 40) | (a[:, 1] << 20) | a[:, 2]  # noqa: E731
        keys = np.sort(key(c))
        has = lambda a: keys[np.clip(np.searchsorted(keys, key(a)), 0, len(keys) - 1)] == key(a)  # noqa: E731
        offs = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
        line_any = np.stack([has(c + np.array(offs[g])) | has(c - np.array(offs[g])) for g in range(13)], 1)      # [V, 13]: row has an entry on mirror line g
        P = sum(int(has(c + np.array(o)).sum()) for o in offs)
        out = []
        for G in (1, 2, 3, 4, 7, 13):
            groups = [list(range(i, min(13, i + G))) for i in range(0, 13, G)]
            nonempty = np.stack([line_any[:, g].any(1) for g in groups], 1)
            rows = int(nonempty.sum()) + int((~nonempty.any(1)).sum())       # one partial row per non-empty group; isolated rows keep their centre
            out.append(f"{G} lines/group: {rows} ({rows / P:.0%} of P, {2 ** (2 * G) - 1} source patterns)")
        print(f"level {lvl}: P = {P}; partial rows with " + "; ".join(out))
