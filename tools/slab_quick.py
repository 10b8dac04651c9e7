"""Slab (output-stationary) vs pair-major sparse convolution on the benchmark scene's rulebooks: us per conv, active TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import experimental, ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

d = torch.device("cuda:0")
pts, tgt = make_scene(0, 150000, 3000, 300)
maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
g = torch.Generator().manual_seed(0)
# (table, Cin, Cout, number of such convolutions in Res16UNet34C)
cases = [(("same", 0, 5), 288, 32, 1), (("same", 0, 3), 96, 96, 3), (("same", 0, 3), 128, 96, 1), (("same", 1, 3), 32, 32, 4), (("same", 1, 3), 96, 96, 3),
         (("same", 1, 3), 128, 96, 1), (("same", 2, 3), 32, 64, 1), (("same", 2, 3), 64, 64, 5), (("same", 2, 3), 128, 128, 3), (("same", 2, 3), 192, 128, 1),
         (("same", 3, 3), 64, 128, 1), (("same", 3, 3), 128, 128, 7), (("same", 3, 3), 256, 256, 3), (("same", 3, 3), 384, 256, 1),
         (("same", 4, 3), 128, 256, 1), (("same", 4, 3), 256, 256, 11),
         (("down", 0), 32, 32, 1), (("down", 1), 32, 32, 1), (("down", 2), 64, 64, 1), (("down", 3), 128, 128, 1),
         (("up", 3), 256, 256, 1), (("up", 2), 256, 128, 1), (("up", 1), 128, 96, 1), (("up", 0), 96, 96, 1)]
sel = os.environ.get("SLAB_CASES")
if sel:
    cases = [cases[int(i)] for i in sel.split(",")]
tot_pair = tot_slab = tot_best = tot_flops = 0.0
for key, cin, cout, mult in cases:
    tab = maps.conv_table(*key); nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    x = torch.randn(n_in, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    P = int((nbr >= 0).sum())
    t_pair = timeit(lambda: ops.pair_conv(x, w, pairs), 5)
    if experimental.slab_conv_supported(K, cin, cout, M, P):
        t_slab = timeit(lambda: experimental.slab_conv(x, w, nbr, n_pairs=P), 5)
        diff = (experimental.slab_conv(x, w, nbr, n_pairs=P) - ops.pair_conv(x, w, pairs)).abs().max().item()
    else:
        t_slab, diff = float("nan"), float("nan")
    fl = 2.0 * P * cin * cout
    best = min(t_pair, t_slab) if t_slab == t_slab else t_pair
    tot_pair += mult * t_pair; tot_slab += mult * (t_slab if t_slab == t_slab else t_pair); tot_best += mult * best; tot_flops += mult * fl
    print(f"{str(key):18s} {cin:3d}->{cout:3d} x{mult:2d} M={M:6d} P={P:7d} d={P / (K * M):.2f} | pair {t_pair:6.0f} us {fl / t_pair / 1e6:5.1f} TF | "
          f"slab {t_slab:6.0f} us {fl / t_slab / 1e6:5.1f} TF | maxdiff {diff:.1e}")
print(f"U-Net total: pair-major {tot_pair / 1e3:.2f} ms ({tot_flops / tot_pair / 1e6:.1f} TF), slab {tot_slab / 1e3:.2f} ms ({tot_flops / tot_slab / 1e6:.1f} TF), "
      f"best of both {tot_best / 1e3:.2f} ms ({tot_flops / tot_best / 1e6:.1f} TF)")
