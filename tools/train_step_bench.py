"""One-scene training step (forward + loss + backward) at the benchmark shape, for profiling: python tools/train_step_bench.py [points] [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import add_training_targets, make_scene
n_pts = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = torch.device("cuda:0")
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
pts, tgt = make_scene(5, n_pts, 3000 if n_pts > 50000 else 400, 300 if n_pts > 50000 else 50)
tgt = add_training_targets(pts, tgt, n_instances=40 if n_pts > 50000 else 10, seed=2)
pts, tgt = pts.to(d), tgt.to(d)


def step():
    for p in model.parameters():
        p.grad = None
    for k in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
        tgt.__dict__.pop(k, None)
    t0 = time.perf_counter()
    losses = model([pts], [tgt])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    (losses["seg_loss"] + losses["inst_loss"]).backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1)

for _ in range(2):
    step()
ts = [step() for _ in range(steps)]
print("forward+loss ms", round(sum(t[0] for t in ts) / steps, 2), "backward ms", round(sum(t[1] for t in ts) / steps, 2))
