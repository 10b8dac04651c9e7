"""Micro-benchmark of the pair-major convolution on real rulebooks: time per conv, active TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene


def timeit(fn, reps=5):
    """Median-free simple timer: microseconds per call over `reps` calls between two HIP events."""
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
# PAIR_CHAINED=product: the list formats of the evaluation forward (chained lists at sparse.PAIR_CHAIN_LEVELS, round 4)
maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3], chained=os.environ.get("PAIR_CHAINED") == "product")
g = torch.Generator().manual_seed(0)
cases = [(("same", 0, 5), 288, 32), (("same", 0, 3), 96, 96), (("same", 0, 3), 128, 96), (("same", 1, 3), 32, 32), (("same", 1, 3), 96, 96),
         (("same", 2, 3), 64, 64), (("same", 2, 3), 128, 128), (("same", 2, 3), 192, 128), (("same", 3, 3), 128, 128), (("same", 3, 3), 256, 256),
         (("same", 3, 3), 384, 256), (("same", 4, 3), 256, 256), (("down", 0), 32, 32), (("up", 0), 128, 96), (("up", 2), 256, 128)]
sel = os.environ.get("PAIR_CASES")
if sel:
    cases = [cases[int(i)] for i in sel.split(",")]
for key, cin, cout in cases:
    tab = maps.conv_table(*key); nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    x = torch.randn(n_in, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    P = int((nbr >= 0).sum())
    t = timeit(lambda: ops.pair_conv(x, w, pairs), 5)
    line = f"M={M} P={P} tiles={pairs.p_cap // 128} | {t:.0f} us | {2.0 * P * cin * cout / t / 1e6:.1f} TF/s active"
    if os.environ.get("PAIR_MODES") == "1":      # the same convolution on the other paths: per-row lists over ALL offsets, pos-based pass 2
        from segdino3d_amd.ops import PairLists
        plain = ops.pair_lists(nbr, P)                                                     # no centre offset, not direct
        old = PairLists(plain.pos, plain.in_idx, plain.tile_k, plain.p_cap, plain.K, plain.M)   # rlist = None: round-2 path
        t_rl = timeit(lambda: ops.pair_conv(x, w, plain), 5)
        t_old = timeit(lambda: ops.pair_conv(x, w, old), 5)
        what = "direct" if pairs.direct else ("centre kernel" if pairs.center >= 0 else "row lists")
        line += f" ({what}) | all offsets + row-list pass 2: {t_rl:.0f} us | all offsets + pos pass 2 (round 2): {t_old:.0f} us"
    if os.environ.get("PAIR_CHAINED") == "1" and key[0] == "same" and key[2] == 3:      # the same convolution on CHAINED lists (round 4)
        ch = ops.pair_lists(nbr, P, center=ops.PAIR_CHAINED)
        t_ch = timeit(lambda: ops.pair_conv(x, w, ch), 5)
        rows = int(ch.rlist[:, 0].sum())
        line += f" | chained lists: {t_ch:.0f} us ({rows} partial rows = {rows / P:.0%} of the entries)"
    print(key, cin, cout, line)
