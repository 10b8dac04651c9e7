"""ATen operators of the training FORWARD + loss by calling line (TorchDispatchMode: every aten call with the innermost segdino3d_amd frame)."""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import add_training_targets, make_scene
from torch.utils._python_dispatch import TorchDispatchMode
d = torch.device("cuda:0")
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
pts, tgt = make_scene(5, 150000, 3000, 300)
tgt = add_training_targets(pts, tgt, n_instances=40, seed=2)
pts, tgt = pts.to(d), tgt.to(d)
agg = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        fr = "?"
        for f in reversed(traceback.extract_stack()[:-1]):
            if "segdino3d_amd" in f.filename:
                fr = f"{os.path.basename(f.filename)}:{f.lineno}"
                break
        agg[(name, fr)] += 1
        return func(*args, **(kwargs or {}))


def fwd():
    for k in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
        tgt.__dict__.pop(k, None)
    return model([pts], [tgt])

for _ in range(2):
    l = fwd(); (l["seg_loss"] + l["inst_loss"]).backward()
with Log():
    l = fwd()
skip = ("aten.view", "aten.detach", "aten.alias", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.unsqueeze", "aten.slice", "aten.select",
        "aten.expand", "aten.as_strided", "aten.permute", "aten.squeeze", "aten.reshape", "aten.is_", "aten.sym_", "aten.empty", "aten.stride", "aten.size")
n = 0
for (name, fr), c in sorted(agg.items(), key=lambda kv: -kv[1]):
    if any(name.startswith(s) for s in skip):
        continue
    n += c
    if c >= 2:
        print(f"{c:4d}  {name:38s} {fr}")
print("aten calls (views excluded):", n)
