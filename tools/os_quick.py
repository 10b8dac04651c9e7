import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
def timeit(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
d = torch.device("cuda:0")
pts, tgt = make_scene(0, 150000, 3000, 300)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
maps.prepare(same=[(l, 3) for l in range(5)], strides=[])
g = torch.Generator().manual_seed(0)
for lvl, cin, cout in ((4, 256, 256), (3, 256, 256), (3, 128, 128), (2, 128, 128)):
    tab = maps.conv_table("same", lvl, 3); nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    x = torch.randn(M, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    P = int((nbr >= 0).sum())
    tp = timeit(lambda: ops.pair_conv(x, w, pairs))
    to = timeit(lambda: ops.gather_gemm(x, w, nbr=nbr))
    print(f"level {lvl} {cin}->{cout}: M={M} P={P} | pair-major {tp:.0f} us | output-stationary gather_gemm {to:.0f} us")
