"""Host (Python + launch) cost of one forward: a scene so small that every kernel is launch-bound,
timed section by section on one stream with perf_counter (no profiler overhead)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd import ops, sparse, architecture, decoder, backbone_mink
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
pts, tgt = make_scene(0, N, 300, 300)
pts = pts.to(d); tgt = tgt.to(d)
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)
wrap(sparse.SceneMaps, "__init__", "SceneMaps.__init__ (voxelise, 1 sync)")
wrap(sparse.SceneMaps, "prepare", "SceneMaps.prepare (tables, pair lists, 1 sync)")
wrap(model.backbone, "forward_sparse", "backbone.forward_sparse (incl. prepare)")
wrap(model.backbone, "forward_wrapper", "backbone.forward_wrapper (all)")
wrap(model.decoder, "forward", "decoder.forward")
wrap(model, "predict_by_feat", "predict_by_feat (post)")
with torch.no_grad():
    for _ in range(5):
        model([pts], [tgt])
    torch.cuda.synchronize()
    acc.clear()
    R = 20
    t0 = time.perf_counter()
    for _ in range(R):
        model([pts], [tgt])
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
print(f"points {N}: {1e3 * tot / R:.2f} ms per forward (host-bound)")
for k, v in acc.items():
    print(f"  {k}: {1e3 * v / R:.2f} ms")
