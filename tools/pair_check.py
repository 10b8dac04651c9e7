import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
pts, tgt = make_scene(31, n_points=9000, n_superpoints=60, n_query2d=5)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
print("n_vox", maps.n_vox)
g = torch.Generator().manual_seed(0)
cases = [(("same", 0, 5), 32, 32)] + [(("same", l, 3), c, c) for l, c in [(0, 96), (1, 32), (2, 64), (3, 128), (4, 256), (1, 96), (2, 128), (3, 256)]] + \
        [(("down", l), ci, co) for l, ci, co in [(0, 32, 32), (1, 32, 64), (2, 64, 128), (3, 128, 256)]] + \
        [(("up", l), ci, co) for l, ci, co in [(3, 256, 256), (2, 256, 128), (1, 128, 96), (0, 96, 96)]]
for key, cin, cout in cases:
    tab = maps.conv_table(*key); nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    x = torch.randn(n_in, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    ref = ops.gather_gemm(x, w, nbr=nbr, nt=1)
    got = ops.pair_conv(x, w, pairs)
    err = (got - ref).abs().max().item()
    print(key, cin, cout, "M", M, "tiles", pairs.p_cap // 128, "n_real", int(pairs.tile_k[-1]), "err %.2e" % err, "BAD" if err > 1e-3 else "")
# ---- details for one failing case
key, cin, cout = ("same", 1, 3), 32, 32
tab = maps.conv_table(*key); nbr, pairs = tab["nbr"], tab["pairs"]
K, M = nbr.shape
x = torch.randn(M, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
ref = ops.gather_gemm(x, w, nbr=nbr, nt=1)
part = ops._WS3.get(pairs.p_cap * cout * 4, d)
part.view(torch.float32).fill_(7.0)
got = ops.pair_conv(x, w, pairs)
torch.cuda.synchronize()
pf = part.view(torch.float32)[: pairs.p_cap * cout].view(pairs.p_cap, cout)
bad_rows = torch.isnan(got).any(dim=1).nonzero().squeeze(1)
print("nan rows", bad_rows.numel(), "of", M, bad_rows[:10].tolist())
untouched = (pf == 7.0).all(dim=1)
real = pairs.in_idx >= 0
print("real pairs", int(real.sum()), "untouched real rows", int((untouched & real).sum()), "nan part rows", int(torch.isnan(pf).any(dim=1).sum()))
ut = (untouched & real).nonzero().squeeze(1)
print("untouched real pair positions (tile, within):", [(int(p) // 128, int(p) % 128) for p in ut[:12]])
nanp = torch.isnan(pf).any(dim=1).nonzero().squeeze(1)
print("nan part positions:", [(int(p) // 128, int(p) % 128) for p in nanp[:12]], "tile_k there:", [int(pairs.tile_k[int(p) // 128]) for p in nanp[:6]])
