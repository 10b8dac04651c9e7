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
for (lvl, k, cin, cout) in [(0, 3, 96, 96), (0, 3, 128, 96), (1, 3, 96, 96), (1, 3, 32, 32), (2, 3, 128, 128), (2, 3, 64, 64), (0, 5, 288, 32)]:
    nbr = maps.same(lvl, k); K, M = nbr.shape
    x = torch.randn(M, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    res = []
    for nt in (-11, -21, -61, -62):
        if cout != 96 and nt == -13: continue
        ref = ops.gather_gemm(x, w, nbr=nbr, nt=1)
        err = (ops.gather_gemm(x, w, nbr=nbr, nt=nt) - ref).abs().max().item()
        res.append(f"{nt}:{timeit(lambda: ops.gather_gemm(x, w, nbr=nbr, nt=nt), 5):.0f}" + ("(BAD %.1e)" % err if err > 1e-3 and not os.environ.get('SD3D_GG_DBG') else ""))
    print(lvl, k, cin, cout, " ".join(res))
