"""Host-sensitivity probe: busy-wait D microseconds (holding the baton / GIL) before the decoder is issued, in every forward."""
import os, sys, time, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd import architecture
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
pool = [tuple(t.to(d) for t in make_scene(j, 150000, 3000, 300)) for j in range(2)]
orig = architecture.Baseline3D.forward_decoder
delay = [0.0]
def slow(self, *a, **k):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < delay[0]:
        pass
    return orig(self, *a, **k)
architecture.Baseline3D.forward_decoder = slow
r = PipelinedRunner(model, 4)
r.run([pool[i % 2] for i in range(60)])
for us in (0, 500, 1000, 2000, 0):
    delay[0] = us * 1e-6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.run([pool[i % 2] for i in range(160)])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"extra host time {us:5d} us per forward: {160 / dt:.1f} scenes/s")
