"""Which torch (ATen) operators does one eval forward still launch, and from which line of the package?
torch.profiler with Python stacks over one single-stream forward at the benchmark shape."""
import os, sys, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import make_scene
from torch.profiler import profile, ProfilerActivity
d = torch.device("cuda:0")
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=200)).to(d).eval()
pts, tgt = make_scene(0, 150000, 3000, 300)
pts, tgt = pts.to(d), tgt.to(d)
with torch.no_grad():
    for _ in range(3):
        model([pts], [tgt])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model([pts], [tgt])
        torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    frame = next((f for f in ev.stack if "segdino3d_amd" in f), ev.stack[0] if ev.stack else "?")
    agg[(ev.name, frame.split("/root/repo/")[-1] if "/root/repo/" in frame else frame[-80:])] += len(ev.kernels)
total = 0
for (name, frame), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    total += n
    print(f"{n:4d}  {name:28s} {frame}")
print("device launches from ATen operators:", total)
