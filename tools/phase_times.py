"""Where does a scene's time go when four scenes are in flight?  HIP events (no serialisation) around the phases of every forward -
map building, the U-Net (one C call), pooling + decoder, post-processing - in the pipelined runner and alone."""
import os, sys, time, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd import architecture, backbone_mink, sparse, plan
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
pool = [tuple(t.to(d) for t in make_scene(j, 150000, 3000, 300)) for j in range(2)]
marks = []                                                    # (label, start event, end event) appended from all threads


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            return f(*a, **k)
        finally:
            e1.record()
            marks.append((label, e0, e1))
    setattr(obj, name, g)


wrap(sparse.SceneMaps, "__init__", "1 voxelise + levels")
wrap(sparse.SceneMaps, "prepare", "2 neighbour tables + pair lists")
wrap(plan.LayerPlan, "run", "3 U-Net (sd3d_run_layers)")
wrap(model.decoder, "forward", "5 decoder")
wrap(model, "predict_by_feat", "6 post-processing")
wrap(model, "forward", "0 whole forward")


def report(tag, n_scenes, wall):
    torch.cuda.synchronize()
    agg = {}
    for label, e0, e1 in marks:
        agg.setdefault(label, []).append(e0.elapsed_time(e1))
    print(f"== {tag}: {n_scenes / wall:.1f} scenes/s, {1e3 * wall / n_scenes:.2f} ms per scene")
    for label in sorted(agg):
        v = agg[label]
        print(f"   {label:34s} {sum(v) / len(v):7.2f} ms on its stream (n={len(v)})")
    marks.clear()


with torch.no_grad():
    for _ in range(3):
        model([pool[0][0]], [pool[0][1]])
    torch.cuda.synchronize(); marks.clear()
    t0 = time.perf_counter()
    for i in range(24):
        model([pool[i % 2][0]], [pool[i % 2][1]])
    torch.cuda.synchronize()
    report("one scene in flight", 24, time.perf_counter() - t0)
r = PipelinedRunner(model, 4)
r.run([pool[i % 2] for i in range(40)])
torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
r.run([pool[i % 2] for i in range(160)])
torch.cuda.synchronize()
report("four scenes in flight", 160, time.perf_counter() - t0)
