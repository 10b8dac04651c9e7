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


def test_runner_plans_equal_shares_per_stream():
    """`PipelinedRunner.plan_batches`: every scene exactly once, streams within one scene of each other, forwards of near-equal size
    no larger than the batch - the driver's 20 steps on 4 streams x batches of 4 become five scenes per stream as 2 + 3."""
    from segdino3d_amd.dist_eval import PipelinedRunner

    class Plan(PipelinedRunner):
        def __init__(self, n, batch):                      # planning only: no device, no streams
            self.n, self.batch = n, batch
    for n, b, k in [(4, 4, 20), (4, 4, 48), (4, 1, 5), (2, 2, 11), (3, 4, 2), (1, 4, 9), (4, 4, 0)]:
        plan = Plan(n, b).plan_batches(k)
        assert len(plan) == n
        assert sorted(i for st in plan for fwd in st for i in fwd) == list(range(k))
        shares = [sum(len(f) for f in st) for st in plan]
        assert max(shares) - min(shares) <= 1
        for st in plan:
            sizes = [len(f) for f in st]
            assert all(1 <= z <= b for z in sizes) and (not sizes or max(sizes) - min(sizes) <= 1)
    assert [[len(f) for f in st] for st in Plan(4, 4).plan_batches(20)] == [[2, 3]] * 4


def test_dense_plan_code_is_a_host_function():
    """sd3d_dense_plan_code needs no GPU: lock-step kernel from 64 row tiles on, split contraction for few tiles and >= 8 chunks."""
    from segdino3d_amd import ops
    assert ops.dense_code(3000, 256, 3072) == 0 and ops.dense_code(2048, 96, 256) == 0
    assert ops.dense_code(200, 256, 256) == -1 and ops.dense_code(200, 1024, 256) == -1
    assert ops.dense_code(200, 96, 256) == 1                      # 3 chunks of 32: no split
    assert ops.dense_code(340, 256, 3072) == 1 and ops.dense_code(301, 256, 3072) == -1


def _fake_kfd(tmp_path, nodes):
    """nodes: list of (simd_count, domain, location_id)."""
    for i, (simd, dom, loc) in enumerate(nodes):
        d = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ndomain {dom}\nlocation_id {loc}\nname x\n")
    return str(tmp_path)


def test_gpus_are_counted_from_sysfs_without_hip(tmp_path):
    """The launcher parent counts GPUs from the KFD topology: CPU agents (simd_count 0) are skipped, the PCI
    address is decoded from location_id, and the *_VISIBLE_DEVICES filters apply in the runtime's order."""
    from segdino3d_amd.dist_eval import kfd_gpu_bdfs, visible_gpu_bdfs
    root = _fake_kfd(tmp_path, [(0, 0, 0), (0, 0, 0), (1024, 0, 0x0500), (1024, 0, 0x1500), (1024, 1, 0x8508), (1024, 0, 0xa500)])
    assert kfd_gpu_bdfs(root) == ["0000:05:00.0", "0000:15:00.0", "0001:85:01.0", "0000:a5:00.0"]
    assert visible_gpu_bdfs(root, env={}) == kfd_gpu_bdfs(root)
    assert visible_gpu_bdfs(root, env={"HIP_VISIBLE_DEVICES": "2,0"}) == ["0001:85:01.0", "0000:05:00.0"]
    assert visible_gpu_bdfs(root, env={"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "0,2"}) == ["0000:15:00.0", "0000:a5:00.0"]
    assert visible_gpu_bdfs(root, env={"CUDA_VISIBLE_DEVICES": "1"}) == ["0000:15:00.0"]
    assert visible_gpu_bdfs(root, env={"HIP_VISIBLE_DEVICES": ""}) == []
    assert visible_gpu_bdfs(root, env={"HIP_VISIBLE_DEVICES": "0,9,1"}) == ["0000:05:00.0"]          # stops at the first invalid index
    assert visible_gpu_bdfs(root, env={"ROCR_VISIBLE_DEVICES": "GPU-abcdef"}) == kfd_gpu_bdfs(root)   # UUIDs are not interpreted
    assert kfd_gpu_bdfs(str(tmp_path / "nothing")) == []


def test_gpus_without_an_accessible_render_node_are_not_counted(tmp_path):
    """A container that was given only some /dev/dri/renderD* nodes still sees every GPU of the host in sysfs: a GPU whose render node
    cannot be opened read-write is left out of the count and of the index -> PCI address list NUMA pinning uses."""
    from segdino3d_amd.dist_eval import kfd_gpu_bdfs, visible_gpu_bdfs
    root = _fake_kfd(tmp_path, [(0, 0, 0), (1024, 0, 0x0500), (1024, 0, 0x1500), (1024, 0, 0x2500)])
    nodes = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for i, minor in ((1, 128), (2, 129), (3, 130)):
        with open(nodes / str(i) / "properties", "a") as f:
            f.write(f"drm_render_minor {minor}\n")
    dev = tmp_path / "dev"
    (dev / "dri").mkdir(parents=True)
    (dev / "dri" / "renderD128").write_text("")
    (dev / "dri" / "renderD130").write_text("")
    assert kfd_gpu_bdfs(root, str(dev)) == ["0000:05:00.0", "0000:25:00.0"]
    assert visible_gpu_bdfs(root, env={"HIP_VISIBLE_DEVICES": "1"}, dev_root=str(dev)) == ["0000:25:00.0"]
    assert kfd_gpu_bdfs(root, str(tmp_path / "no_dev")) == []


def test_bench_parent_does_not_import_a_gpu_count_from_torch():
    """`launch_ranks` / `pin_rank_to_cores` in bench.py must not call torch.cuda.* (the parent stays GPU-free; a rank only touches ITS GPU)."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for fn in tree.body:
        if isinstance(fn, ast.FunctionDef) and fn.name in ("launch_ranks", "pin_rank_to_cores"):
            body = ast.get_source_segment(src, fn)
            assert "torch.cuda" not in body.split('"""')[2], fn.name


def test_launch_ranks_takes_every_rank_down_when_one_fails():
    """Rank 3 of 8 exits non-zero while the others sit in a (simulated) collective: the parent notices within a poll interval,
    terminates the other seven, relays what rank 0 printed, and returns rank 3's code (bench.py:243)."""
    import time
    from segdino3d_amd.dist_eval import launch_ranks
    sleeper = "import sys, time; print('rank0 line', flush=True); time.sleep(120)"
    cmds = []
    for r in range(8):
        code = "import sys, time; time.sleep(0.5); sys.exit(7)" if r == 3 else sleeper
        cmds.append(([sys.executable, "-c", code], dict(os.environ)))
    sink, t0 = [], time.monotonic()
    rc = launch_ranks(cmds, sink, poll_s=0.05)
    dt = time.monotonic() - t0
    assert rc == 7
    assert dt < 30, dt                                            # not the sleepers' 120 s
    assert b"rank0 line" in b"".join(sink)


def test_launch_ranks_success_and_grace_period():
    import time
    from segdino3d_amd.dist_eval import launch_ranks
    ok = [([sys.executable, "-c", f"print('{{\"n_gpus\": 2}}') if {r} == 0 else None"], dict(os.environ)) for r in range(2)]
    sink = []
    assert launch_ranks(ok, sink, poll_s=0.05) == 0 and b"n_gpus" in b"".join(sink)
    # rank 1 hangs after rank 0 finished: killed after the grace period, reported as 124
    hang = [([sys.executable, "-c", "pass"], dict(os.environ)), ([sys.executable, "-c", "import time; time.sleep(120)"], dict(os.environ))]
    t0 = time.monotonic()
    assert launch_ranks(hang, [], poll_s=0.05, grace_s=1.0) == 124
    assert time.monotonic() - t0 < 30
