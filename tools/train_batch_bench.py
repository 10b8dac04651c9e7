"""Training step on a batch of scenes (the reference trains 4 scenes per rank, train_engine_3d.py:88): python tools/train_batch_bench.py [scenes] [points]"""
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import add_training_targets, make_scene
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
d = torch.device("cuda:0")
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
batch = []
for b in range(B):
    pts, tgt = make_scene(60 + b, n_pts, 3000, 300)
    tgt = add_training_targets(pts, tgt, n_instances=40, seed=b)
    batch.append((pts.to(d), tgt.to(d)))


def step():
    for p in model.parameters():
        p.grad = None
    for _, t in batch:
        for k in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
            t.__dict__.pop(k, None)
    losses = model([p for p, _ in batch], [t for _, t in batch])
    (losses["seg_loss"] + losses["inst_loss"]).backward()
    return losses

for _ in range(2):
    step()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
R = 3
for _ in range(R):
    l = step()
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / R
out = dict(scenes_per_step=B, points_per_scene=n_pts, ms_per_step=round(ms, 1), ms_per_scene=round(ms / B, 1), scenes_per_second=round(1e3 * B / ms, 2),
           peak_torch_memory_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), seg_loss=round(float(l["seg_loss"]), 4),
           inst_loss=round(float(l["inst_loss"]), 4))
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/train_batch_{B}.json", "w"), indent=1)
