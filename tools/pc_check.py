"""LAB BUILD (tools/ablate.sh build; run with SD3D_LIB=_ab/libsegdino3d_hip_ablate.so): producer / consumer pass 1 against the lock-step / weight-stationary kernels on the benchmark scene's rulebooks: same bits?  us per
convolution (incl. pass 2) with each, alone on the GPU and with the several-scenes-in-flight hint.  usage: python tools/pc_check.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import _lib, ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene

lib = _lib.load()
import ctypes
lib.sd3d_set_pair_pc.restype, lib.sd3d_set_pair_pc.argtypes = ctypes.c_int, [ctypes.c_int]      # lab-build symbol, not in the product header


def timeit(fn, reps=7):
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
maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3], chained=True)
g = torch.Generator().manual_seed(0)
cases = [(("same", 0, 5), 288, 32), (("same", 0, 3), 96, 96), (("same", 0, 3), 128, 96), (("same", 1, 3), 32, 32), (("same", 1, 3), 96, 96),
         (("same", 2, 3), 64, 64), (("same", 2, 3), 128, 128), (("same", 2, 3), 192, 128), (("same", 3, 3), 128, 128), (("same", 3, 3), 256, 256),
         (("same", 3, 3), 384, 256), (("same", 4, 3), 256, 256), (("down", 0), 32, 32), (("up", 0), 128, 96), (("up", 2), 256, 128)]
sel = os.environ.get("PAIR_CASES")
if sel:
    cases = [cases[int(i)] for i in sel.split(",")]
print("| table | Cin -> Cout | P | default us | pc us | same bits | crowded: default us | pc us | same bits |")
print("|---|---|---:|---:|---:|---|---:|---:|---|")
tot = [0.0, 0.0, 0.0, 0.0]
for key, cin, cout in cases:
    tab = maps.conv_table(*key); nbr, pairs = tab["nbr"], tab["pairs"]
    K, M = nbr.shape
    n_in = int(nbr.max().item()) + 1
    x = torch.randn(n_in, cin, generator=g).to(d); w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
    res = torch.randn(M, cout, generator=g).to(d)
    scale = (0.5 + torch.rand(cout, generator=g)).to(d); shift = (0.1 * torch.randn(cout, generator=g)).to(d)
    P = int((nbr >= 0).sum())
    run = lambda: ops.pair_conv(x, w, pairs, scale=scale, shift=shift, res=res, act="relu")   # noqa: E731
    row = []
    for crowded in (1, 4):
        with ops.scenes_in_flight(crowded):
            lib.sd3d_set_pair_pc(0)
            ref = run().clone(); t0 = timeit(run)
            lib.sd3d_set_pair_pc(2)
            got = run().clone(); t1 = timeit(run)
            lib.sd3d_set_pair_pc(0)
        row += [t0, t1, torch.equal(ref, got)]
    tot[0] += row[0]; tot[1] += row[1]; tot[2] += row[3]; tot[3] += row[4]
    print(f"| {key} | {cin} -> {cout} | {P} | {row[0]:.0f} | {row[1]:.0f} | {row[2]} | {row[3]:.0f} | {row[4]:.0f} | {row[5]} |")
print(f"| sum | | | {tot[0]:.0f} | {tot[1]:.0f} | | {tot[2]:.0f} | {tot[3]:.0f} | |")
