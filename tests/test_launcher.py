"""Host logic of the self-launching `bench.py --gpus N` (CPU only): the per-rank command / environment construction
(what `python -m torch.distributed.launch --nproc_per_node N` sets up in the reference's `scripts/eval.sh:12-19`), the
NUMA-aware host-core placement, and the parent's refusal to run N ranks on fewer GPUs."""
import os
import subprocess
import sys

from segdino3d_amd.dist_eval import cores_for_rank, gpu_numa_node, parse_cpulist, rank_commands

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_commands_environment():
    cmds = rank_commands("/x/bench.py", ["--gpus", "4", "--steps", "8"], 4, 29511, python="/usr/bin/python3", base_env={"A": "1"})
    assert len(cmds) == 4
    for r, (cmd, env) in enumerate(cmds):
        assert cmd == ["/usr/bin/python3", "/x/bench.py", "--gpus", "4", "--steps", "8"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29511"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["A"] == "1"
    assert cmds[0][1] is not cmds[1][1]


def test_parse_cpulist():
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("") == []


def _fake_sysfs(tmp_path, gpu_nodes, node_cpus):
    for bdf, node in gpu_nodes.items():
        d = tmp_path / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpus in node_cpus.items():
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    return str(tmp_path)


def test_numa_placement(tmp_path):
    root = _fake_sysfs(tmp_path, {"0000:05:00.0": 0, "0000:85:00.0": 1}, {0: "0-15,32-47", 1: "16-31,48-63"})
    assert gpu_numa_node("0000:05:00.0", root) == 0
    assert gpu_numa_node("0000:85:00.0", root) == 1
    assert gpu_numa_node("0000:99:00.0", root) == -1
    allowed = list(range(64))
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]                      # 8 GPUs, 4 per socket
    slices = [cores_for_rank(r, nodes, allowed, root) for r in range(8)]
    assert all(len(s) == 8 for s in slices)
    assert sorted(c for s in slices for c in s) == allowed              # disjoint, complete
    node0 = set(parse_cpulist("0-15,32-47"))
    assert all(set(s) <= node0 for s in slices[:4]) and all(not (set(s) & node0) for s in slices[4:])
    # a restricted affinity mask is honoured
    slices = [cores_for_rank(r, nodes, list(range(0, 24)), root) for r in range(8)]
    assert all(set(s) <= set(range(24)) and s for s in slices)
    assert all(set(s) <= node0 for s in slices[:4])


def test_numa_unknown_falls_back_to_even_slices(tmp_path):
    root = _fake_sysfs(tmp_path, {}, {})
    slices = [cores_for_rank(r, [-1, -1, -1, -1], list(range(16)), root) for r in range(4)]
    assert slices == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    # mixed: rank 1's node is unknown -> it gets what the placed ranks leave
    root = _fake_sysfs(tmp_path / "b", {}, {0: "0-7"})
    assert cores_for_rank(0, [0, -1], list(range(16)), root) == list(range(8))
    assert cores_for_rank(1, [0, -1], list(range(16)), root) == list(range(8, 16))


def test_bench_parent_refuses_more_ranks_than_gpus():
    """No GPU in the CPU container: `--gpus 2` without a launcher must fail loudly in the parent, never print a 1-GPU line."""
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("SD3D_SHARE_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "--gpus 2" in r.stderr


def test_bench_rank_refuses_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "n_gpus" not in r.stdout
