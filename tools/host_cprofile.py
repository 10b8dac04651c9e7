"""cProfile of the host side of one forward (tiny scene, launch-bound): where the Python time goes."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
pts, tgt = make_scene(0, 3000, 300, 300)
pts, tgt = pts.to(d), tgt.to(d)
with torch.no_grad():
    for _ in range(5):
        model([pts], [tgt])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        model([pts], [tgt])
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
