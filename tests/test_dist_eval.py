"""world_size-2 `gloo` test of the scene sharding + per-scene record all-gather (the N > 1 path of
bench.py / the evaluation driver).  CPU only."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_scenes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from segdino3d_amd.dist_eval import all_gather_records, run_sharded, shard_scenes
    mine = shard_scenes(n_scenes, rank, world)
    assert mine == list(range(rank, n_scenes, world))
    table = run_sharded(n_scenes, lambda i: (i, 1000 + i, 2.5 * i, rank))
    torch.save(table, os.path.join(out_dir, f"table{rank}.pt"))
    # ragged + empty contribution
    local = torch.arange(3 * (rank * 2), dtype=torch.float64).reshape(rank * 2, 3)      # rank 0: 0 rows, rank 1: 2 rows
    parts = all_gather_records(local)
    assert [p.shape[0] for p in parts] == [0, 2]
    dist.barrier()
    dist.destroy_process_group()


def test_scene_sharding_and_record_gather(tmp_path):
    world, n_scenes = 2, 7
    mp.spawn(_worker, args=(world, _free_port(), n_scenes, str(tmp_path)), nprocs=world, join=True)
    t0 = torch.load(tmp_path / "table0.pt")
    t1 = torch.load(tmp_path / "table1.pt")
    assert torch.equal(t0, t1)
    assert t0.shape == (n_scenes, 4)
    assert t0[:, 0].tolist() == list(range(n_scenes))
    assert t0[:, 1].tolist() == [1000 + i for i in range(n_scenes)]
    assert t0[:, 3].tolist() == [i % world for i in range(n_scenes)]


def test_single_process_passthrough():
    from segdino3d_amd.dist_eval import all_gather_records, merge_by_scene
    r = torch.tensor([[2.0, 5.0], [0.0, 1.0]], dtype=torch.float64)
    parts = all_gather_records(r)
    assert len(parts) == 1 and torch.equal(merge_by_scene(parts)[:, 0], torch.tensor([0.0, 2.0], dtype=torch.float64))
