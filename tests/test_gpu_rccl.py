"""The RCCL (`backend="nccl"`) path on the one GPU a box has: world size 1, in a FRESH process.

SURVEY 8(e): scenes shard over ranks with no data-path collective; one all-gather of per-scene records closes a run.  The gloo tests
(`test_dist_eval.py`, `test_eval_ap.py`) cover the logic at world size 2 / 8 on the CPU; these run the same functions on DEVICE tensors
through RCCL - communicator init with `device_id=`, all_gather, all_reduce, barrier, teardown - so that the first multi-GPU run has only
the topology left to discover.  The children are started by the GPU-free spawn server of `tests/conftest.py` (never a fork + exec of
this process, which has initialised the GPU).  Reference: `segdino3d/utils/dist_utils.py:197-246` (`nccl` at :233)."""
import json
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_env(rank=0, world=1):
    env = dict(os.environ)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SD3D_DIST_BACKEND", None)                          # the default: nccl
    return env


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


def test_record_gathers_over_rccl_at_world_size_one(fresh_process):
    r = fresh_process([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py")], env=_rank_env(), timeout=600)
    assert r["rc"] == 0, r["stderr"][-4000:]
    out = _last_json(r["stdout"])
    assert out["ok"] and out["backend"] == "nccl" and out["world"] == 1
    assert out["records"] == [3] and out["run_sharded_rows"] == 7
    print("RCCL world size 1:", out)


def test_bench_runs_as_a_rank_on_the_nccl_backend(fresh_process):
    """`bench.py --gpus 1` under a launcher's environment: process group on nccl, both barriers, the max-over-ranks all-reduce and the
    closing all-gather of the per-rank records execute (the `dist is not None` branches)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
           "--no-end-to-end", "--preroll-seconds", "0.3", "--sustain-seconds", "0.2"]
    r = fresh_process(cmd, env=_rank_env(), timeout=900)
    assert r["rc"] == 0, r["stderr"][-4000:]
    line = _last_json(r["stdout"])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["value"] > 0
    pg = line["process_group"]
    assert pg is not None and pg["backend"] == "nccl" and pg["world_size"] == 1
    assert len(line["per_rank_records"]) == 1 and line["per_rank_records"][0][0] == 0.0
    print("bench.py as rank 0 of 1 on nccl:", line["value"], "scenes/s")
