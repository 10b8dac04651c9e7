"""One RCCL rank, run as a FRESH process by tests/test_gpu_rccl.py (through the GPU-free spawn server): the `backend="nccl"` path of
`segdino3d_amd.dist_eval` - what `bench.py --gpus N` and the sharded evaluation driver use - at the world size the environment names
(1 on the one-GPU box: every collective still goes through RCCL's init, communicator and kernels; only the topology is left to a
multi-GPU node).  Mirrors the reference's process-group setup (`segdino3d/utils/dist_utils.py:197-246`, backend `nccl` at :233).
Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    dev = torch.device("cuda", local if os.environ.get("SD3D_SHARE_GPU") != "1" else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rank": dist.get_rank()}
    from segdino3d_amd import dist_eval, eval_ap
    from test_eval_ap import _records

    # 1. fixed-width records on DEVICE tensors: counts gather + padded payload gather (ragged: rank r contributes 2 r + 3 rows)
    n_local = 2 * rank + 3
    recs = (torch.arange(n_local * 4, dtype=torch.float64, device=dev).reshape(n_local, 4) + 1000.0 * rank)
    parts = dist_eval.all_gather_records(recs)
    assert len(parts) == world and all(p.is_cuda for p in parts)
    assert [p.shape[0] for p in parts] == [2 * r + 3 for r in range(world)]
    assert torch.equal(parts[rank], recs)
    empty = dist_eval.all_gather_records(torch.zeros((0, 4), dtype=torch.float64, device=dev))
    assert [p.shape[0] for p in empty] == [0] * world
    out["records"] = [int(p.shape[0]) for p in parts]

    # 2. run_sharded: scene sharding + width all-reduce(MAX) + gather + merge, everything on the device
    n_scenes = 7
    table = dist_eval.run_sharded(n_scenes, lambda i: (i, 1000 + i, 2.5 * i, rank), device=dev)
    assert table.is_cuda and table.shape == (n_scenes, 4)
    assert table[:, 0].tolist() == list(range(n_scenes)) and table[:, 1].tolist() == [1000 + i for i in range(n_scenes)]
    assert table[:, 3].tolist() == [i % world for i in range(n_scenes)]
    out["run_sharded_rows"] = int(table.shape[0])

    # 3. the AP records of the evaluation driver through the device: same AP as the single-process evaluation
    z, class_labels, valid, groups, opts, id_to_label, preds, gts, scene_recs = _records(None)
    mine = [(i, scene_recs[i]) for i in dist_eval.shard_scenes(len(scene_recs), rank, world)]
    allrecs = dist_eval.all_gather_ap_records(mine, device=dev)
    assert [sid for sid, _ in allrecs] == list(range(len(scene_recs)))
    for (sid, got), ref in zip(allrecs, scene_recs):
        for f in ref.__dataclass_fields__:
            assert np.array_equal(getattr(got, f), getattr(ref, f)), (sid, f)
    ap, pr_rc = eval_ap.evaluate_records([r for _, r in allrecs], class_labels, valid, opts)
    metrics = eval_ap.compute_averages(ap, pr_rc, opts, class_labels)
    ref_ap = dict(zip((str(k) for k in z["default_keys"]), z["default_vals"]))["all_ap"]
    assert abs(metrics["all_ap"] - ref_ap) < 1e-12, (metrics["all_ap"], ref_ap)
    out["all_ap"] = float(metrics["all_ap"])

    # 4. the collectives bench.py closes a run with: barrier, all_reduce(MAX) of the timed span, all_gather of one record per rank
    dist.barrier()
    tt = torch.tensor([1.0 + rank], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt.item()) == float(world)
    rec = torch.tensor([float(rank), 150000.0, 138690.0, 7.9], dtype=torch.float64, device=dev)
    gathered = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(gathered, rec)
    assert torch.stack(gathered)[:, 0].tolist() == [float(r) for r in range(world)]
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    out["ok"] = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
