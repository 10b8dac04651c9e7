"""How evenly do output-row slabs split a convolution's work?  pairs / padded 16-row MFMA slots per slab: mean, max."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
pts, tgt = make_scene(0, 150000, 3000, 300)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
for lvl, R in ((0, 288), (1, 288), (2, 224), (2, 112), (3, 112), (3, 56), (4, 112), (4, 32)):
    nbr = maps.same(lvl, 3)
    K, M = nbr.shape
    ns = (M + R - 1) // R
    has = torch.zeros(K, ns * R, dtype=torch.int32, device=d)
    has[:, :M] = (nbr >= 0).int()
    per = has.view(K, ns, R).sum(2)                       # pairs per (offset, slab)
    pairs = per.sum(0).float()
    slots = (((per + 15) // 16) * 16).sum(0).float()      # 16-row MFMA granules incl. padding
    print(f"level {lvl} R={R:3d}: {ns:4d} slabs | pairs/slab mean {pairs.mean():7.0f} max {pairs.max():6.0f} (max/mean {pairs.max() / pairs.mean():.2f}) | "
          f"padded slots mean {slots.mean():7.0f} max {slots.max():6.0f} (max/mean {slots.max() / slots.mean():.2f}) | fill {pairs.sum() / slots.sum():.2f} | "
          f"time ~ max slots / mean pairs = {slots.max() / pairs.mean():.2f}x ideal")
