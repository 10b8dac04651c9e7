"""f-2 measurement (lives under tests/ because it times the oracle): per-scene AP association on the device vs the
oracle (numpy) on the host, full-size scene
(600 predictions x 150 k points, ~60 ground-truth instances of a 198-class label set)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import eval_ap, _lib, ops
from oracle import eval_ref as E
d = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
N, n, n_inst = 150_000, 600, 60
valid = tuple(range(2, 200)); class_labels = tuple(f"c{i}" for i in valid); id_to_label = dict(zip(valid, class_labels))
owner = torch.randint(0, n_inst, (N,), generator=gen)
sem = torch.randint(0, 200, (n_inst,), generator=gen)
gt = torch.where(torch.isin(sem[owner], torch.tensor(valid)), sem[owner] * 1000 + owner + 1, owner + 1)
# predictions: each covers most of one instance plus noise (like a trained model's masks: a few % of the points each)
masks = torch.zeros(n, N, dtype=torch.bool)
for p in range(n):
    o = int(torch.randint(0, n_inst, (1,), generator=gen))
    masks[p] = ((owner == o) & (torch.rand(N, generator=gen) > 0.2)) | (torch.rand(N, generator=gen) > 0.995)
labels = torch.tensor([max(0, min(len(valid) - 1, int(sem[int(torch.randint(0, n_inst, (1,), generator=gen))]) - 2)) for _ in range(n)])
scores = torch.rand(n, generator=gen)
opts = eval_ap.get_options(None)
md, ld, sd_, gd = masks.to(d), labels.to(d), scores.to(d), gt.to(d)
for _ in range(3):
    rec = eval_ap.assign_scene(md, ld, sd_, gd, opts, valid)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 20
for _ in range(R):
    rec = eval_ap.assign_scene(md, ld, sd_, gd, opts, valid)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / R
# kernel alone (HIP events)
lib = _lib.load()
gt_index = torch.zeros(N, dtype=torch.int32, device=d)
counts = torch.empty(n, 64, dtype=torch.int32, device=d)
m8 = md.view(torch.uint8)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(R):
    lib.sd3d_mask_overlaps(m8.data_ptr(), m8.stride(0), n, gt_index.data_ptr(), N, 64, counts.data_ptr(), ops._stream())
e1.record(); torch.cuda.synchronize()
t_k = e0.elapsed_time(e1) / R * 1e-3
# host oracle
pred_info = {f"0_{i}": dict(mask=masks[i].numpy(), label_id=valid[int(labels[i])], conf=float(scores[i])) for i in range(n)}
gt_np = gt.numpy()
t0 = time.perf_counter()
g2p, p2g = E.assign_instances(pred_info, gt_np, opts, valid, class_labels, id_to_label)
t_cpu = time.perf_counter() - t0
print(json.dumps({"workload": f"{n} predictions x {N} points, {len(rec.gt_id)} gt instances, mask density {float(masks.float().mean()):.3f}",
                  "device_assign_scene_ms": round(1e3 * t_dev, 3), "mask_overlaps_kernel_us": round(1e6 * t_k, 1),
                  "kernel_GBps_of_mask_bytes": round(n * N / t_k / 1e9, 1), "oracle_numpy_1core_ms": round(1e3 * t_cpu, 1),
                  "pairs": int(len(rec.pair_pred))}))
