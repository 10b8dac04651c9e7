"""Where does the host time of one forward go?  cProfile over single-stream forwards (GPU box)."""
import os, sys, cProfile, pstats, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
pts, tgt = make_scene(0, N, 3000 if N > 10000 else 300, 300)
pts = pts.to(d); tgt = tgt.to(d)
with torch.no_grad():
    for _ in range(3):
        model([pts], [tgt])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); c0 = time.thread_time()
    for _ in range(10):
        model([pts], [tgt])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0; c1 = time.thread_time() - c0
    print(f"10 forwards: issue {t_issue*100:.2f} ms/scene, wall {t1*100:.2f} ms/scene, thread cpu {c1*100:.2f} ms/scene")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        model([pts], [tgt])
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
