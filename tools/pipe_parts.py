"""Pipelined throughput of parts of the forward (what would trimming the non-GEMM work buy?): pipe_parts.py <streams> <scenes>"""
import os, sys, time, copy, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd import ops
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 48
pool = []
for j in range(2):
    pts, tgt = make_scene(j, 150000, 3000, 300)
    pool.append((pts.to(d), tgt.to(d)))
scenes = lambda n: [(pool[i % 2][0], copy.copy(pool[i % 2][1])) for i in range(n)]


class Part(torch.nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, samples, targets):
        return self.fn(samples, targets)


def backbone_only(samples, targets):
    with ops.stream_scope():
        return model.forward_backbone(samples, targets)


def unet_only(samples, targets):                          # maps + U-Net, no voxel features / pooling
    with ops.stream_scope():
        bb = model.backbone
        pts = samples[0]
        maps = SceneMaps(pts, bb.voxel_size, 5, superpoints=targets[0]["extra_features"]["super_point_masks"])
        vf = maps.voxel_features(pts, targets[0]["extra_features"]["points_2dfeats"], 0, 288)
        return bb.forward_sparse(maps, vf)

def maps_only(samples, targets):
    with ops.stream_scope():
        maps = SceneMaps(samples[0], 0.02, 5, superpoints=targets[0]["extra_features"]["super_point_masks"])
        maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
        return maps


def maps_nosp(samples, targets):
    with ops.stream_scope():
        maps = SceneMaps(samples[0], 0.02, 5)
        maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
        return maps


def unet_pool(samples, targets):
    with ops.stream_scope():
        bb = model.backbone
        pts = samples[0]
        maps = SceneMaps(pts, bb.voxel_size, 5, superpoints=targets[0]["extra_features"]["super_point_masks"])
        vf = maps.voxel_features(pts, targets[0]["extra_features"]["points_2dfeats"], 0, 288)
        x = bb.forward_sparse(maps, vf)
        return maps.pool(x, 96)

for name, fn in (("full forward", None), ("backbone wrapper only", backbone_only), ("maps + voxel features + U-Net + pool", unet_pool),
                 ("maps + voxel features + U-Net", unet_only), ("maps (with superpoint sort) + pair lists", maps_only),
                 ("maps without superpoints + pair lists", maps_nosp)):
    runner = PipelinedRunner(model if fn is None else Part(fn), NS, d)
    runner.run(scenes(2 * NS)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run(scenes(R)); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {1e3 * dt / R:.2f} ms/scene, {R / dt:.1f} scenes/s")
