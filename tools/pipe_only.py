"""Only the pipelined runner (for profiling its steady state): pipe_only.py <streams> <scenes>"""
import os, sys, time, copy, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pool = []
for j in range(2):
    pts, tgt = make_scene(j, 150000, 3000, 300)
    pool.append((pts.to(d), tgt.to(d)))
runner = PipelinedRunner(model, NS, d)
scenes = lambda n: [(pool[i % 2][0], copy.copy(pool[i % 2][1])) for i in range(n)]
runner.run(scenes(2 * NS)); torch.cuda.synchronize()
t0 = time.perf_counter()
runner.run(scenes(R)); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"streams {NS}: {1e3 * dt / R:.2f} ms/scene, {R / dt:.1f} scenes/s")
