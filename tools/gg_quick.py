import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
from tools.bench_gg import timeit
d = torch.device("cuda:0")
pts, tgt = make_scene(0, 150000, 3000, 300)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
g = torch.Generator().manual_seed(0)
for (lvl, k, cin, cout) in [(0, 3, 96, 96), (1, 3, 96, 96), (2, 3, 128, 128), (0, 5, 288, 32)]:
    nbr = maps.same(lvl, k); K, M = nbr.shape
    x = torch.randn(M, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    res = []
    for nt in (-11, -21, -22, -31):
        if cout == 32 and nt == -22: continue
        res.append(f"{nt}:{timeit(lambda: ops.gather_gemm(x, w, nbr=nbr, nt=nt), 5):.0f}")
    print(lvl, k, cin, cout, " ".join(res))
