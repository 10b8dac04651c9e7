"""Which torch (ATen) operators does one training step (forward + loss + backward) launch, and from where?  torch.profiler with Python
stacks over one step at the benchmark shape; backward-pass operators are attributed to the autograd node that ran them."""
import os, sys, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import add_training_targets, make_scene
from torch.profiler import profile, ProfilerActivity
d = torch.device("cuda:0")
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
pts, tgt = make_scene(5, 150000, 3000, 300)
tgt = add_training_targets(pts, tgt, n_instances=40, seed=2)
pts, tgt = pts.to(d), tgt.to(d)


def step():
    for p in model.parameters():
        p.grad = None
    for k in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
        tgt.__dict__.pop(k, None)
    losses = model([pts], [tgt])
    (losses["seg_loss"] + losses["inst_loss"]).backward()
    torch.cuda.synchronize()


for _ in range(2):
    step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or not ev.kernels:
        continue
    frame = next((f for f in ev.stack if "segdino3d_amd" in f), None)
    if frame is None:
        frame = next((f for f in ev.stack if "autograd" in f or "Backward" in f), ev.stack[0] if ev.stack else "?")
    agg[(ev.name, frame.split("/root/repo/")[-1] if "/root/repo/" in frame else frame[-90:])] += len(ev.kernels)
total = 0
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}  {name:28s} {frame}")
for (name, frame), n in agg.items():
    total += n
print("device launches from ATen operators:", total)
