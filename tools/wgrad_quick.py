"""Weight-gradient kernel on real rulebooks: time per call, TFLOP/s of real flops."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops, train_ops
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
out = []
for key, cin, cout in [(("same", 0, 5), 288, 32), (("same", 0, 3), 96, 96), (("same", 1, 3), 96, 96), (("same", 2, 3), 128, 128), (("same", 3, 3), 256, 256),
                       (("same", 4, 3), 256, 256), (("same", 1, 3), 32, 32), (("up", 2), 256, 128), (("same", 2, 3), 64, 64), (("up", 0), 128, 96), (("same", 2, 3), 192, 128), (("same", 0, 3), 128, 96)]:
    tab = maps.conv_table(*key); nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    x = torch.randn(n_in, cin, generator=g).to(d); dy = torch.randn(M, cout, generator=g).to(d)
    P = int((pairs.in_idx >= 0).sum())
    t = timeit(lambda: train_ops.pair_wgrad(dy, x, pairs))
    out.append(f"{key} {cin}->{cout}: {t:.0f} us {2.0 * P * cin * cout / t / 1e6:.1f} TF")
# decoder-sized Linear gradients (identity lists)
for M, cin, cout in [(2441, 256, 256), (2441, 256, 1024), (3000, 256, 3072), (200, 256, 256)]:
    nbr = torch.arange(M, dtype=torch.int32, device=d).unsqueeze(0).contiguous()
    pairs = ops.pair_lists(nbr, M)
    x = torch.randn(M, cin, generator=g).to(d); dy = torch.randn(M, cout, generator=g).to(d)
    t = timeit(lambda: train_ops.pair_wgrad(dy, x, pairs))
    out.append(f"linear M={M} {cin}->{cout}: {t:.0f} us {2.0 * M * cin * cout / t / 1e6:.1f} TF")
print("\n".join(out))
