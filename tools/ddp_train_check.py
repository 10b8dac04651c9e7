"""Training step under DistributedDataParallel (the reference's multi-GPU training: train_engine_3d.py wraps the model in DDP and
all-reduces ~200 MB of fp32 gradients per step).  Two ranks, one scene each; checks that the gradients DDP leaves on every rank
equal the mean of the two ranks' single-process gradients, i.e. that the autograd nodes over HIP kernels cooperate with DDP's
bucketed all-reduce hooks.  On a one-GPU box both ranks share cuda:0 and the process group is gloo (RCCL refuses two ranks on one
device); on a multi-GPU node export SD3D_DDP_BACKEND=nccl and each rank takes its own device.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_train_check.py"""
import json, os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from segdino3d_amd.configs import scannet200_model_cfg
from segdino3d_amd.synth import add_training_targets, make_scene

backend = os.environ.get("SD3D_DDP_BACKEND", "gloo")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend, rank=rank, world_size=world)
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=-1)).to(dev).train()
pts, tgt = make_scene(30 + rank, n_points=12000, n_superpoints=120, n_query2d=20)
tgt = add_training_targets(pts, tgt, n_instances=6, seed=rank)
pts, tgt = pts.to(dev), tgt.to(dev)


def step(m):
    for p in m.parameters():
        p.grad = None
    for k in ("query_inst_sem_masks", "instance_centers", "instance_sizes"):
        tgt.__dict__.pop(k, None)
    torch.manual_seed(100 + rank)                       # the query subset of this rank's scene
    losses = m([pts], [tgt])
    (losses["seg_loss"] + losses["inst_loss"]).backward()
    return {k: float(v.detach()) for k, v in losses.items()}

# 1. single-process gradients of this rank's scene, averaged over the ranks by hand
l_local = step(model)
names = [n for n, p in model.named_parameters()]
local = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
mean = {}
for n in names:
    g = local[n].clone()
    dist.all_reduce(g)
    mean[n] = g / world
# 2. the same step under DDP
ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=True, bucket_cap_mb=25)
t0 = time.perf_counter()
l_ddp = step(ddp)
torch.cuda.synchronize()
t_ddp = time.perf_counter() - t0
worst, n_bytes = 0.0, 0
for n, p in model.named_parameters():
    g = p.grad if p.grad is not None else torch.zeros_like(p)
    n_bytes += g.numel() * 4
    ref = mean[n]
    scale = max(float(ref.abs().max()), 1e-12)
    worst = max(worst, float((g - ref).abs().max()) / scale if float(ref.abs().max()) > 1e-9 else float(g.abs().max()))
# every rank must hold the same gradients
chk = torch.stack([p.grad.double().sum() for p in model.parameters() if p.grad is not None]).sum().reshape(1)
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
ok = worst < 1e-5 and all(float(b) == float(both[0]) for b in both) and abs(l_ddp["inst_loss"] - l_local["inst_loss"]) < 1e-6 * abs(l_local["inst_loss"])
if rank == 0:
    out = dict(ok=bool(ok), world=world, backend=backend, worst_relative_gradient_difference=worst, gradient_bytes_all_reduced=n_bytes,
               parameters=len(names), losses_rank0=l_ddp, ddp_step_seconds_first_call=round(t_ddp, 3))
    print(json.dumps(out))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/ddp_train_check.json", "w"), indent=1)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
