"""Superpoint pooling alone on the benchmark scene (and a 500 k-point one): us per launch and a checksum of the pooled rows (same-bits
check between two builds of the library: SD3D_LIB=...).  usage: python tools/pool_quick.py"""
import hashlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene

d = torch.device("cuda:0")
for n, sp in ((150000, 3000), (500000, 10000)):
    pts, tgt = make_scene(0, n, sp, 300)
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(maps.n_vox[0], 96, generator=g).to(d)
    f, pos = maps.pool(feat, 96)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        maps.pool(feat, 96)
    e0.record()
    for _ in range(20):
        maps.pool(feat, 96)
    e1.record()
    torch.cuda.synchronize()
    h = hashlib.sha256(f.cpu().numpy().tobytes() + pos.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{n} points / {maps.n_superpoints} superpoints: {1e3 * e0.elapsed_time(e1) / 20:.1f} us per pool call (all its launches), checksum {h}")
