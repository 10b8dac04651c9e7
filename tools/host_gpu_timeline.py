"""One scene in flight: where is the HOST and where is the GPU, phase by phase?  perf_counter at entry / return of each phase and HIP
events around it (no profiler), both relative to the forward's start.  A phase whose GPU start trails its host entry by more than
the queue depth is GPU-bound there; one whose GPU start follows its host entry closely is host-bound (the GPU waits for launches)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd import sparse, plan
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(int(os.environ.get("QUERY_NUM", "200")), d)
pool = [tuple(t.to(d) for t in make_scene(j, 150000, 3000, 300)) for j in range(2)]
marks = []


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        e0.record()
        try:
            return f(*a, **k)
        finally:
            e1.record()
            marks.append((label, h0, time.perf_counter(), e0, e1))
    setattr(obj, name, g)


wrap(sparse.SceneMaps, "__init__", "1 voxelise + levels (sync 1)")
wrap(sparse.SceneMaps, "prepare", "2 maps + stem lists")
wrap(sparse.SceneMaps, "voxel_features", "2b voxel_mean")
wrap(plan.LayerPlan, "run", "3 U-Net (run_layers, forks)")
wrap(sparse.SceneMaps, "pool", "4 pooling")
wrap(model.decoder, "forward", "5 decoder")
wrap(model, "predict_by_feat", "6 post-processing (sync 2)")
with torch.no_grad():
    for i in range(6):
        model([pool[i % 2][0]], [pool[i % 2][1]])
    torch.cuda.synchronize()
    rows = {}
    n = 10
    for i in range(n):
        marks.clear()
        torch.cuda.synchronize()
        base = torch.cuda.Event(enable_timing=True)
        hb = time.perf_counter()
        base.record()
        model([pool[i % 2][0]], [pool[i % 2][1]])
        torch.cuda.synchronize()
        hend = time.perf_counter()
        for label, h0, h1, e0, e1 in marks:
            r = rows.setdefault(label, [0.0] * 4)
            r[0] += (h0 - hb) * 1e3; r[1] += (h1 - hb) * 1e3; r[2] += base.elapsed_time(e0); r[3] += base.elapsed_time(e1)
        rows.setdefault("0 whole forward (host return = GPU done)", [0.0] * 4)[1] += (hend - hb) * 1e3
print("phase | host enters | host returns | GPU starts | GPU ends   (ms after the forward's start, mean of %d forwards)" % n)
for label in sorted(rows):
    r = [v / n for v in rows[label]]
    print(f"{label:45s} {r[0]:7.3f} {r[1]:7.3f} {r[2]:7.3f} {r[3]:7.3f}")
