"""Kernel maps and pair lists of one benchmark scene, timed in isolation (HIP events, nothing else on the GPU): what the builders cost when
they do not share the chip with the convolutions (in the forward they run on the side stream, and a kernel trace reports their durations
inflated by whatever runs beside them)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops, sparse
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
pts, tgt = make_scene(0, 150000, 3000, 300)
pts = pts.to(d)
sp = tgt.extra_features["super_point_masks"].to(d)


def timed(f, n=30):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print(f"voxelise + levels + read-back (SceneMaps.__init__): {timed(lambda: SceneMaps(pts, 0.02, 5, superpoints=sp)):8.1f} us")
maps = SceneMaps(pts, 0.02, 5, superpoints=sp)
L = 5
print("voxels per level", maps.n_vox)


def all_maps():
    return ops.kernel_maps_hier(maps.keys, maps.parents, maps.n_vox, sparse.offsets_device(3, maps.order, d), sparse.offsets_device(5, maps.order, d),
                                sparse.inv27_table(maps.order), None, perm8=maps._perm8)


print(f"all kernel maps of the scene (sd3d_kernel_maps_hier): {timed(all_maps):8.1f} us")
nbr3, nbr5, strides = all_maps()
worst = lambda t: t.shape[0] * t.shape[1]
LEAN = os.environ.get("LISTS_LEAN", "1") != "0"          # the evaluation forward's tables: no position table, unused capacity unwritten
chained = [(nbr3[l], worst(nbr3[l]), ops.PAIR_CHAINED, False, LEAN) for l in range(3)]
plain3 = [(nbr3[l], worst(nbr3[l]), -1, False, LEAN) for l in (3, 4)]
stem = [(nbr5, worst(nbr5), -1, False, LEAN)]
updown = []
for l, (dn, up) in enumerate(strides):
    updown += [(dn, maps.n_vox[l], -1, False, LEAN), (up, maps.n_vox[l], -1, True, LEAN)]
print("lean tables" if LEAN else "tables with position table and filled capacity (training; rounds 1 - 4)")
print(f"chained lists of levels 0-2:                            {timed(lambda: ops.pair_lists_batch(chained)):8.1f} us")
print(f"stem 5^3 list:                                           {timed(lambda: ops.pair_lists_batch(stem)):8.1f} us")
print(f"plain 3^3 lists of levels 3-4 + 8 stride-2 lists:      {timed(lambda: ops.pair_lists_batch(plain3 + updown)):8.1f} us")
print(f"all 14 tables in one call:                              {timed(lambda: ops.pair_lists_batch(chained + stem + plain3 + updown)):8.1f} us")
