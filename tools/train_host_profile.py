"""cProfile of one training step's backward (host side) at the benchmark shape: where the Python time goes.  usage: python tools/train_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import add_training_targets, make_scene

d = torch.device("cuda:0")
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(d).train()
pts, tgt = make_scene(5, 150000, 3000, 300)
tgt = add_training_targets(pts, tgt, n_instances=40, seed=2)
pts, tgt = pts.to(d), tgt.to(d)


def fwd():
    for p in model.parameters():
        p.grad = None
    for k in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
        tgt.__dict__.pop(k, None)
    losses = model([pts], [tgt])
    return losses["seg_loss"] + losses["inst_loss"]


for _ in range(3):
    fwd().backward()
torch.cuda.synchronize()
loss = fwd()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
loss.backward()
pr.disable()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"backward: host returns after {1e3 * t_issue:.1f} ms, GPU done after {1e3 * (time.perf_counter() - t0):.1f} ms")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
pr = cProfile.Profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr.enable()
loss = fwd()
pr.disable()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"forward + loss: host returns after {1e3 * t_issue:.1f} ms, GPU done after {1e3 * (time.perf_counter() - t0):.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
