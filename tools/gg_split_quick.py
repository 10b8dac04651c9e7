"""Micro-benchmark: exact fp32 lock-step gather-GEMM vs the split-bf16 variants on real rulebooks."""
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
print("lvl k cin cout M | fp32(auto) fp32(lds) compact4 | x3 x6 | TF/s dense-equivalent of x3")
for (lvl, k, cin, cout) in [(0, 3, 32, 32), (0, 3, 96, 96), (0, 3, 128, 96), (1, 3, 64, 64), (1, 3, 96, 96), (2, 3, 128, 128), (2, 3, 256, 128),
                            (3, 3, 256, 256), (3, 3, 384, 256), (4, 3, 256, 256), (0, 5, 288, 32)]:
    nbr = maps.same(lvl, k); K, M = nbr.shape
    x = torch.randn(M, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    s3, s6 = ops.split_weights(w, 3), ops.split_weights(w, 6)
    ref = ops.gather_gemm(x, w, nbr=nbr, nt=1)
    t = {}
    t["auto"] = timeit(lambda: ops.gather_gemm(x, w, nbr=nbr), 5)
    t["lds"] = timeit(lambda: ops.gather_gemm(x, w, nbr=nbr, nt=-10 - min(4, cout // 32)), 5)
    t["c4"] = timeit(lambda: ops.gather_gemm(x, w, nbr=nbr, nt=-61), 5)
    t["x3"] = timeit(lambda: ops.gather_gemm(x, w, nbr=nbr, wt_split=s3), 5)
    t["x6"] = timeit(lambda: ops.gather_gemm(x, w, nbr=nbr, wt_split=s6), 5)
    e3 = (ops.gather_gemm(x, w, nbr=nbr, wt_split=s3) - ref).abs().max().item()
    e6 = (ops.gather_gemm(x, w, nbr=nbr, wt_split=s6) - ref).abs().max().item()
    dens = float((nbr >= 0).float().mean())
    tf = 2.0 * K * M * cin * cout / (t["x3"] * 1e-6) / 1e12
    print(lvl, k, cin, cout, M, f"dens={dens:.2f} |", " ".join(f"{n}:{v:.0f}" for n, v in t.items()), f"| err3={e3:.1e} err6={e6:.1e} | {tf:.0f}")
