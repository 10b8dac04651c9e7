"""Per-section wall time of the forward under the 3-thread pipelined runner (sums over threads / scenes)."""
import os, sys, time, copy, threading, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd import ops, sparse
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pts, tgt = make_scene(0, 150000, 3000, 300)
pts = pts.to(d); tgt = tgt.to(d)
acc = {}
lock = threading.Lock()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            with lock:
                acc[label] = acc.get(label, 0.0) + dt
    setattr(obj, name, g)
wrap(sparse.SceneMaps, "__init__", "SceneMaps.__init__ (voxelise, sync 1)")
wrap(sparse.SceneMaps, "prepare", "SceneMaps.prepare (tables, sync 2)")
wrap(model.backbone, "forward_sparse", "backbone.forward_sparse (incl. prepare)")
wrap(model.backbone, "forward_wrapper", "backbone.forward_wrapper (all)")
wrap(model.decoder, "forward", "decoder.forward (no sync)")
wrap(model, "predict_by_feat", "predict_by_feat (post, final sync)")
runner = PipelinedRunner(model, NS, d)
scenes = lambda n: [(pts, copy.copy(tgt)) for _ in range(n)]
runner.run(scenes(6))
torch.cuda.synchronize(); acc.clear()
R = 30
t0 = time.perf_counter()
runner.run(scenes(R))
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"streams {NS}: {1e3 * tot / R:.2f} ms per scene")
for k, v in acc.items():
    print(f"  {k}: {1e3 * v / R:.2f} ms per scene (thread wall)")
