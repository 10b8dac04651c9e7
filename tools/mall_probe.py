"""Does the pair-major convolution get cheaper per pair when its partial products fit the 256 MB Infinity Cache?  The same layer
on the first 1/2, 1/4 of the output rows (own pair lists): us per 1000 pairs."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

d = torch.device("cuda:0")
pts, tgt = make_scene(0, 150000, 3000, 300)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
g = torch.Generator().manual_seed(0)
for (lvl, cin, cout) in ((2, 128, 128), (1, 96, 96), (3, 256, 256)):
    nbr = maps.same(lvl, 3)
    K, M = nbr.shape
    x = torch.randn(M, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    for frac in (1, 2, 4):
        m = M // frac
        sub = nbr[:, :m].contiguous()
        P = int((sub >= 0).sum())
        pl = ops.pair_lists(sub, P)
        t = timeit(lambda: ops.pair_conv(x, w, pl))
        print(f"level {lvl} {cin}->{cout} rows {m:6d} pairs {P:7d} partials {P * cout * 4 / 1e6:6.1f} MB: {t:6.1f} us = {1e3 * t / P:.3f} us / 1000 pairs, {2.0 * P * cin * cout / t / 1e6:.1f} TF")
