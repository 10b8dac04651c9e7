import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
pts, tgt = make_scene(31, n_points=9000, n_superpoints=60, n_query2d=5)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
maps.prepare(same=[(l, 3) for l in range(5)], strides=[0])
tab = maps.conv_table("same", 1, 3); nbr, pairs = tab["nbr"], tab["pairs"]
K, M = nbr.shape
g = torch.Generator().manual_seed(0)
cin, cout = 64, 64
x = torch.randn(M, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
part = ops._WS3.get(pairs.p_cap * cout * 4, d)
got = ops.pair_conv(x, w, pairs)
torch.cuda.synchronize()
pf = part.view(torch.float32)[: pairs.p_cap * cout].view(pairs.p_cap, cout).clone()
# expected partial products
idx = pairs.in_idx.long()
tk = pairs.tile_k.long()[:-1].repeat_interleave(128)
valid = idx >= 0
exp = torch.zeros_like(pf)
xs = x[idx.clamp(min=0)]
for k in range(K):
    m = valid & (tk == k)
    if m.any():
        exp[m] = xs[m] @ w[k].T
bad = ((pf - exp).abs() > 1e-3) | torch.isnan(pf)
bad = bad & valid[:, None]
print("bad elements", int(bad.sum()), "bad rows", int(bad.any(dim=1).sum()), "of valid", int(valid.sum()))
rows = bad.any(dim=1).nonzero().squeeze(1)
import collections
print("bad rows: lane-in-subtile histogram", sorted(collections.Counter((rows % 32).tolist()).items())[:40])
print("bad rows: subtile-in-tile histogram", sorted(collections.Counter(((rows // 32) % 4).tolist()).items()))
print("bad cols histogram", sorted(collections.Counter(bad.nonzero()[:, 1].tolist()).items())[:70])
print("bad tiles", sorted(collections.Counter((rows // 128).tolist()).items())[:40])
print("n_real", int(pairs.tile_k[-1]), "tile_k", pairs.tile_k[:40].tolist())
# ---- what is in the first tile?
t = 0
sl = slice(t * 128, t * 128 + 128)
v = valid[sl]
P = pf[sl][v]; X = xs[sl][v]
print("tile", t, "valid", int(v.sum()), "k", int(pairs.tile_k[t]))
for kk in range(min(K, 6)):
    print("  vs W[%d]: max err %.3e" % (kk, (P - X @ w[kk].T).abs().max().item()))
print("  |P| max %.3e, P[0,:8] %s" % (P.abs().max().item(), P[0, :8].tolist()))
E = X @ w[int(pairs.tile_k[t])].T
print("  E[0,:8]", E[0, :8].tolist())
# per-chunk partial sums: maybe only part of the chunks were accumulated
for c0 in range(0, cin, 32):
    Ec = X[:, c0:c0 + 32] @ w[int(pairs.tile_k[t])][:, c0:c0 + 32].T
    print("  only chunk %d: max err %.3e" % (c0 // 32, (P - Ec).abs().max().item()))
print("  ratio P/E first row", (P[0, :8] / E[0, :8]).tolist())
