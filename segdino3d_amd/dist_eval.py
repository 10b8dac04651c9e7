"""Scene-sharded multi-GPU evaluation support.

The reference evaluates on a single GPU ("Not support multi-card evaluation.",
`evaluation/evaluate_3d.py:45`).  Scenes are independent units (batch size 1, BatchNorm running stats),
so here rank r simply takes scenes r, r + W, r + 2W, ... with NO collective on the data path; the only
exchange is one all-gather of per-scene records at the end (throughput records, or the compact output
of the AP matching).  Records have a variable count per rank, so the exchange is length-then-padded
payload, the same pattern as the reference's unused `all_gather` helper
(`segdino3d/utils/dist_utils.py:148-194`), but on fixed-width numeric rows instead of pickles.
Backend "nccl" is RCCL on ROCm (one process per GPU, xGMI); "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import os

import torch
import torch.distributed as dist


def shard_scenes(n_scenes: int, rank: int, world: int) -> List[int]:
    """Round-robin scene -> rank assignment (scene i on rank i mod W)."""
    return list(range(rank, n_scenes, world))


def all_gather_records(records: torch.Tensor, group=None) -> List[torch.Tensor]:
    """records [n_local, width] (same width and dtype on every rank) -> list of per-rank tensors.

    Two collectives: an all-gather of the row counts, then one all-gather of the payload padded to the
    largest count.  Works with device tensors on "nccl" and CPU tensors on "gloo"."""
    if not (dist.is_available() and dist.is_initialized()):
        return [records]
    world = dist.get_world_size(group)
    if records.dim() != 2:
        raise ValueError("records must be [n, width]")
    n_local = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1)
    padded = records.new_zeros((n_max, records.shape[1]))
    padded[: records.shape[0]] = records
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded.contiguous(), group=group)
    return [g[:c] for g, c in zip(gathered, counts)]


def merge_by_scene(per_rank: Sequence[torch.Tensor], key_col: int = 0) -> torch.Tensor:
    """Concatenate the gathered rows and order them by the scene-id column."""
    allrows = torch.cat([t for t in per_rank if t.numel() > 0]) if any(t.numel() for t in per_rank) else per_rank[0]
    if allrows.numel() == 0:
        return allrows
    order = torch.argsort(allrows[:, key_col], stable=True)
    return allrows[order]


def run_sharded(n_scenes: int, process_scene: Callable[[int], Sequence[float]], device="cpu") -> torch.Tensor:
    """Each rank runs `process_scene(i)` for its scenes (returning a fixed-width numeric record whose
    first entry is the scene id); returns the merged [n_scenes, width] table on every rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    rows = [list(process_scene(i)) for i in shard_scenes(n_scenes, rank, world)]
    width = len(rows[0]) if rows else 0
    if dist.is_initialized():                      # ranks with no scene still need the common width
        w = torch.tensor([width], dtype=torch.int64, device=device)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        width = int(w.item())
    local = torch.tensor(rows, dtype=torch.float64, device=device).reshape(len(rows), width)
    return merge_by_scene(all_gather_records(local))


# ---- one process per GPU: launch line and host-core placement ------------------------------------------------------------
def rank_commands(script: str, argv: Sequence[str], n_ranks: int, port: int, python: str = None, base_env=None):
    """The N child processes of a self-launched run: [(command, environment)] for rank 0..N-1, one per GPU of ONE node - what
    `python -m torch.distributed.launch --nproc_per_node N` of the reference's launch lines (`scripts/eval.sh:12-19`,
    `scripts/train.sh:21-28`) sets up, without the launcher in between: RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve).  The PARENT that calls this must not have touched
    the GPU (the children are fresh processes, never an exec of an initialised one)."""
    import sys
    python = python or sys.executable
    base = dict(os.environ if base_env is None else base_env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: RCCL across processes needs it on this driver
    out = []
    for r in range(int(n_ranks)):
        env = dict(base)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out.append(([python, script] + list(argv), env))
    return out


def kfd_gpu_bdfs(sysfs_root: str = "/sys", dev_root: str = "/dev") -> List[str]:
    """PCI addresses ('dddd:bb:dd.f') of the GPU agents of this node in KFD order, read from
    `<sysfs>/class/kfd/kfd/topology/nodes/<i>/properties` - no HIP / HSA call, so a launcher parent that uses it stays
    provably GPU-free (`torch.cuda.device_count()` may open /dev/kfd on a ROCm build without amdsmi).  A node is a GPU when its
    `simd_count` is non-zero (CPU agents report 0); `location_id` = bus << 8 | device << 3 | function, `domain` the PCI domain.
    sysfs lists every GPU of the HOST even inside a container that was handed only some `/dev/dri/renderD*` nodes: a GPU whose render
    node (`drm_render_minor`) this process cannot open read-write is not counted - the runtime would not enumerate it either, and a
    rank started for it would fail at `set_device` after its peers have entered `init_process_group`."""
    base = os.path.join(sysfs_root, "class", "kfd", "kfd", "topology", "nodes")
    try:
        ids = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for i in ids:
        props = {}
        try:
            with open(os.path.join(base, str(i), "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    if v.strip().lstrip("-").isdigit():
                        props[k] = int(v)
        except OSError:
            continue
        if props.get("simd_count", 0) <= 0:
            continue
        minor = props.get("drm_render_minor", 0)
        if minor > 0 and not os.access(os.path.join(dev_root, "dri", f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        loc, dom = props.get("location_id", 0), props.get("domain", 0)
        out.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    return out


def visible_gpu_bdfs(sysfs_root: str = "/sys", env=None, dev_root: str = "/dev") -> List[str]:
    """`kfd_gpu_bdfs` filtered the way the runtime will filter the devices of a child process: ROCR_VISIBLE_DEVICES first (it
    renumbers the agents), then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES on the renumbered list.  Only integer lists are
    interpreted (UUID entries leave the list as it is); an empty string hides every GPU, as in the runtime."""
    env = os.environ if env is None else env
    gpus = kfd_gpu_bdfs(sysfs_root, dev_root)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        val = env.get(var)
        if val is None and var == "HIP_VISIBLE_DEVICES":
            val = env.get("CUDA_VISIBLE_DEVICES")
        if val is None:
            continue
        parts = [p.strip() for p in val.split(",") if p.strip() != ""]
        if val.strip() == "":
            return []
        if not all(p.lstrip("-").isdigit() for p in parts):
            continue
        picked = []
        for p in parts:
            j = int(p)
            if j < 0 or j >= len(gpus):
                break                                          # the runtime stops at the first invalid index
            picked.append(gpus[j])
        gpus = picked
    return gpus


def launch_ranks(commands, rank0_stdout_sink, poll_s: float = 0.2, grace_s: float = 600.0, popen=None):
    """Run the per-rank (command, environment) list of `rank_commands` as child processes and supervise ALL of them: rank 0's
    stdout is drained by a reader thread into `rank0_stdout_sink` (a list of byte chunks; the JSON line), every child is polled, and the
    FIRST non-zero exit of ANY rank terminates the others at once (a rank that dies early would otherwise leave its peers in
    `init_process_group` / a barrier until the collective timeout).  After rank 0 has finished cleanly the others get `grace_s`
    more seconds (they only have the closing barrier and teardown left).  Returns the exit code: 0, the first failing rank's
    code, or 124 for a rank killed after the grace period."""
    import subprocess
    import sys
    import threading
    import time
    popen = popen or subprocess.Popen
    procs = []
    for rank, (cmd, env) in enumerate(commands):
        procs.append(popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr))

    def drain():
        while True:
            chunk = procs[0].stdout.read(65536)
            if not chunk:
                break
            rank0_stdout_sink.append(chunk)
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    rc, t_rank0_done = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                rc = bad[0][1]
                print(f"launch_ranks: rank {bad[0][0]} exited with code {rc}; terminating the other ranks", file=sys.stderr)
                break
            if all(c == 0 for c in codes):
                break
            if codes[0] == 0:
                t_rank0_done = t_rank0_done or time.monotonic()
                if time.monotonic() - t_rank0_done > grace_s:
                    rc = 124
                    print(f"launch_ranks: ranks {[r for r, c in enumerate(codes) if c is None]} still running {grace_s:.0f} s after rank 0 "
                          "finished; killing them", file=sys.stderr)
                    break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        reader.join(timeout=10)
    return rc


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    cores = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cores += list(range(int(lo), int(hi or lo) + 1))
    return cores


def gpu_numa_node(pci_bus_id: str, sysfs_root: str = "/sys") -> int:
    """NUMA node a GPU hangs off, from `<sysfs>/bus/pci/devices/<domain:bus:dev.fn>/numa_node` (the same file the DRM node
    `/sys/class/drm/card*/device/numa_node` links to); -1 when the platform does not say."""
    for name in (pci_bus_id.lower(), pci_bus_id.upper()):
        path = os.path.join(sysfs_root, "bus", "pci", "devices", name, "numa_node")
        try:
            with open(path) as f:
                return int(f.read().strip())
        except (OSError, ValueError):
            continue
    return -1


def cores_for_rank(local_rank: int, gpu_nodes: Sequence[int], allowed: Sequence[int], sysfs_root: str = "/sys") -> List[int]:
    """Host cores for the issuing threads of `local_rank`: the cores of ITS GPU's NUMA node (gpu_nodes[r] = node of the GPU of
    local rank r) that this process may run on, divided evenly among the ranks whose GPUs share that node; ranks whose node is
    unknown (-1) - or whose node has no allowed core - share an even slice of whatever the NUMA-placed ranks leave."""
    allowed = sorted(allowed)
    n = len(gpu_nodes)
    node_cores = {}
    for node in set(gpu_nodes):
        if node < 0:
            continue
        try:
            with open(os.path.join(sysfs_root, "devices", "system", "node", f"node{node}", "cpulist")) as f:
                cs = [c for c in parse_cpulist(f.read()) if c in set(allowed)]
        except (OSError, ValueError):
            cs = []
        node_cores[node] = cs
    placed = [r for r in range(n) if node_cores.get(gpu_nodes[r])]
    mine = gpu_nodes[local_rank]
    if local_rank in placed:
        peers = [r for r in placed if gpu_nodes[r] == mine]
        cs = node_cores[mine]
        per = max(1, len(cs) // len(peers))
        i = peers.index(local_rank)
        return cs[i * per:(i + 1) * per] or cs
    taken = set(c for r in placed for c in node_cores[gpu_nodes[r]])
    rest = [c for c in allowed if c not in taken] or allowed
    loose = [r for r in range(n) if r not in placed]
    per = max(1, len(rest) // len(loose))
    i = loose.index(local_rank)
    return rest[i * per:(i + 1) * per] or rest


class PipelinedRunner:
    """Keeps `n_streams` scenes in flight on ONE GPU: each worker thread owns a HIP stream and runs whole
    eval forwards on it.  A single forward is a chain of ~500 dependent launches, many of them far too
    small to fill 256 CUs (decoder Linears on 200 queries, the stride-16 U-Net level, radix-sort passes),
    plus host synchronisations; another scene's kernels fill those holes.  ctypes and torch release the GIL while they launch / wait, so two Python threads
    are enough to keep both streams fed.  Results keep their submission order.
    """

    def __init__(self, model, n_streams: int = 2, device=None, batch: int = 1):
        """batch > 1: every forward takes `batch` consecutive scenes as ONE block-diagonal sparse tensor
        (sparse.BatchSceneMaps; evaluation only) - each stream then keeps a whole batch in flight."""
        import threading
        self.model = model
        self.n = max(1, int(n_streams))
        self.batch = max(1, int(batch))
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
        self._threading = threading
        self.switch_interval = float(os.environ.get("SD3D_SWITCH_INTERVAL", "2e-4"))
        # one issuing thread at a time, handed over while a thread waits for the GPU (ops.wait_event)
        self.use_baton = os.environ.get("SD3D_BATON", "1") != "0"
        self._keep = True

    def plan_batches(self, n_scenes: int):
        """Which scenes each stream runs, in which forwards: scene i goes to stream i mod n (every stream gets the same number of
        scenes +- 1, whatever the batch size), and a stream cuts ITS scenes into ceil(share / batch) forwards of near-equal size -
        so a short list (the driver's 20 steps on 4 streams x batches of 4: five scenes per stream, forwards of 3 + 2) keeps all
        streams busy to the end instead of leaving one stream a whole extra batch.  -> [[scene ids of a forward, ...] per stream]"""
        plan = []
        forced = getattr(self, "forward_sizes", None)            # e.g. [2, 3]: every stream cuts its scenes into forwards of these sizes, cyclically
        for w in range(self.n):
            mine = list(range(w, n_scenes, self.n))
            if forced:
                cuts, k = [0], 0
                while cuts[-1] < len(mine):
                    cuts.append(min(len(mine), cuts[-1] + max(1, int(forced[k % len(forced)]))))
                    k += 1
                plan.append([mine[cuts[k]:cuts[k + 1]] for k in range(len(cuts) - 1)])
                continue
            nb = (len(mine) + self.batch - 1) // self.batch
            cuts = [len(mine) * k // nb for k in range(nb + 1)] if nb else [0]
            plan.append([mine[cuts[k]:cuts[k + 1]] for k in range(nb)])
        return plan

    def run(self, scenes, on_result=None, keep=True):
        """scenes: sequence of (points, target) already on the device, or an iterator that yields them lazily
        (e.g. io_scene.ScenePrefetcher: each worker pulls its next scene when it is ready for it).  Returns the list of
        model outputs in submission order.  keep=False: an output is dropped as soon as `on_result` has seen it (a long
        evaluation with host-resident outputs would otherwise pin ~100 MB per scene until the run returns)."""
        from . import ops
        self._keep = bool(keep)
        if not hasattr(scenes, "__getitem__"):
            with ops.scenes_in_flight(self.n):
                return self._run_stream(scenes, on_result)
        with ops.scenes_in_flight(self.n):
            return self._run_list(scenes, on_result)

    def _run_list(self, scenes, on_result=None):
        results = [None] * len(scenes)
        errors = []

        baton = self._threading.Lock() if (self.n > 1 and self.use_baton) else None

        def work(wid):
            from . import ops
            try:
                torch.cuda.set_device(self.device)
                if baton is not None:
                    baton.acquire()
                    ops.set_baton(baton)
                with torch.cuda.stream(self.streams[wid]), torch.no_grad():
                    for ids in self.plan_batches(len(scenes))[wid]:
                        out = self.model([scenes[i][0] for i in ids], [scenes[i][1] for i in ids])
                        for j, i in enumerate(ids):
                            results[i] = [out[j]]
                            if on_result is not None:
                                on_result(i, results[i])
                            if not self._keep:
                                results[i] = None
                        del out
                    ops.wait_event(ops.stream_event())
            except BaseException as e:  # noqa: BLE001 - re-raised in the caller's thread
                errors.append(e)
            finally:
                if baton is not None:
                    ops.set_baton(None)
                    baton.release()

        if self.n == 1:
            work(0)
        else:
            import sys
            main = torch.cuda.current_stream(self.device)
            for s in self.streams:
                s.wait_stream(main)
            threads = [self._threading.Thread(target=work, args=(w,)) for w in range(self.n)]
            # a worker that wakes from a host sync must not wait a whole 5 ms GIL slice behind its siblings
            prev_switch = sys.getswitchinterval()
            sys.setswitchinterval(self.switch_interval)
            try:
                for t in threads:
                    t.start()
                for t in threads:
                    t.join()
            finally:
                sys.setswitchinterval(prev_switch)
            for s in self.streams:
                main.wait_stream(s)
        if errors:
            raise errors[0]
        return results


def all_gather_ap_records(local, device="cpu"):
    """local: sequence of (scene_id, eval_ap.SceneRecord) evaluated on this rank -> list of (scene_id, SceneRecord)
    of ALL ranks, ordered by scene id, on every rank.  Each record travels as one float64 row
    [scene_id, payload length, payload..., zero padding] (eval_ap.SceneRecord.pack); ~50 KB per scene, the
    [n_pred, N] masks never leave their GPU."""
    import numpy as np
    from .eval_ap import SceneRecord
    rows = []
    for sid, rec in local:
        payload = rec.pack()
        rows.append(np.concatenate([np.array([float(sid), float(len(payload))]), payload]))
    width = max((len(r) for r in rows), default=2)
    if dist.is_available() and dist.is_initialized():
        w = torch.tensor([width], dtype=torch.int64, device=device)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        width = int(w.item())
    table = np.zeros((len(rows), width), dtype=np.float64)
    for i, r in enumerate(rows):
        table[i, :len(r)] = r
    merged = merge_by_scene(all_gather_records(torch.from_numpy(table).to(device)))
    out = []
    for row in merged.cpu().numpy():
        n = int(row[1])
        out.append((int(row[0]), SceneRecord.unpack(row[2:2 + n])))
    return out


def _pipelined_run_stream(self, it, on_result=None):
    """Workers share one iterator (guarded by a lock; the prefetcher is itself thread-safe and ordered)."""
    results, errors = {}, []
    it = iter(it)
    take = self._threading.Lock()
    counter = [0]
    baton = self._threading.Lock() if (self.n > 1 and self.use_baton) else None

    def work(wid):
        from . import ops
        try:
            torch.cuda.set_device(self.device)
            if baton is not None:
                baton.acquire()
                ops.set_baton(baton)
            with torch.cuda.stream(self.streams[wid]), torch.no_grad():
                while True:
                    group = []
                    with take:
                        for _ in range(self.batch):
                            try:
                                pts, tgt = next(it)
                            except StopIteration:
                                break
                            group.append((counter[0], pts, tgt))
                            counter[0] += 1
                    if not group:
                        break
                    out = self.model([g[1] for g in group], [g[2] for g in group])
                    for j, (i, _, _) in enumerate(group):
                        results[i] = [out[j]]
                        if on_result is not None:
                            on_result(i, results[i])
                        if not self._keep:
                            results[i] = None
                    del out, group
                ops.wait_event(ops.stream_event())
        except BaseException as e:  # noqa: BLE001 - re-raised in the caller's thread
            errors.append(e)
        finally:
            if baton is not None:
                ops.set_baton(None)
                baton.release()

    main = torch.cuda.current_stream(self.device)
    for s in self.streams:
        s.wait_stream(main)
    threads = [self._threading.Thread(target=work, args=(w,)) for w in range(self.n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s in self.streams:
        main.wait_stream(s)
    if errors:
        raise errors[0]
    return [results[i] for i in range(len(results))]


PipelinedRunner._run_stream = _pipelined_run_stream
