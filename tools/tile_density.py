"""Ceiling of an output-stationary implicit GEMM on the benchmark scene: rows are Morton-ordered, a tile of T consecutive
output rows runs one MFMA row block per offset k that has at least one neighbour among the tile's rows (empty (tile, k)
blocks are skipped).  Efficiency = pairs / (T * non-empty blocks)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
for seed in (0, 21):
    pts, tgt = make_scene(seed, 150000, 3000, 300)
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
    for lvl in range(5):
        nbr = maps.same(lvl, 3)
        K, M = nbr.shape
        line = f"seed {seed} level {lvl}: {M:6d} rows, {int((nbr >= 0).sum()) / M:5.2f} neighbours/row |"
        for T in (8, 16, 32, 64):
            nt = (M + T - 1) // T
            has = torch.zeros(K, nt * T, dtype=torch.int32, device=d)
            has[:, :M] = (nbr >= 0).int()
            per = has.view(K, nt, T).sum(2)
            blocks = int((per > 0).sum())
            line += f" T={T}: blocks/tile {blocks / nt:5.2f} fill {int(per.sum()) / (blocks * T):.2f} |"
        print(line)
