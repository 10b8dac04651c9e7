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
