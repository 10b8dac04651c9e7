#!/usr/bin/env python3
"""Headline benchmark: scenes/sec of the eval-mode SegDINO3D forward on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (N > 1: launched by torch.distributed.run, backend "nccl" = RCCL).  A *step* is one
SCENE through the eval-mode `Baseline3D.forward` (backbone + superpoint pooling + 6-layer decoder + post-processing):
a synthetic ScanNet200-like scene (150 k points, 3000 superpoints, 300 2D queries, `query_num=200`,
fp32) whose inputs are already resident in HBM; outputs stay on the device.  `--batch` scenes share a forward
(one block-diagonal sparse tensor, bit-identical per scene; default 4), `--streams` forwards are in flight.  Scenes are independent
units, so ranks shard them with no data-path collective (weak scaling); one all-gather of per-scene
records (scene id, points, voxels, ms) closes the run, as in the north-star's metric exchange.
Inside a rank the K timed steps are issued by `--streams` host threads, each on its own HIP stream
(segdino3d_amd.dist_eval.PipelinedRunner): a forward is ~500 dependent launches, many too small to
fill 256 CUs, so several scenes in flight raise scenes/s over back-to-back forwards; the
single-stream latency is reported next to it (`single_scene`).
`--gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (one child process per GPU,
segdino3d_amd.dist_eval.rank_commands - the parent never touches the GPU), relays rank 0's JSON line and
exits non-zero if any rank fails; under an external launcher (torch.distributed.run) it is one of the ranks.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the dominant kernel = the pair-major sparse convolution (sd3d_pair_conv: pair_gemm_* + pair_reduce_kernel,
                 55 launches per forward, 95 % of the flops): its active-pair flops 2*P*Cin*Cout (P = measured rulebook
                 size) divided by its HIP-event-measured time against the 157.3 TFLOP/s fp32 matrix peak (the bound in exact
                 fp32: 53 flop per algorithmic byte), its algorithmic bytes and the PMC-measured HBM-side bytes
  single_scene : the same forward with ONE scene in flight (SURVEY 8(d) defines the metric at batch = 1 scene per GPU):
                 latency and scenes/s, next to `value` = scenes/s with `--streams` scenes in flight per GPU
  cpu_baseline : the oracle's whole forward timed on the host cores (rank 0, N = 1 only)
"""
from __future__ import annotations

import os as _os
# HIP maps streams onto 4 hardware queues by default; scenes in flight on streams that share a queue block each
# other head-of-line (measured: 4 streams 85 -> 102 scenes/s with 8 queues).  Must be set before the runtime starts.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak


class GemmTimer:
    """HIP-event pair around every gather_gemm launch on the current stream."""

    def __init__(self):
        self.records = []
        self._cur = None
        self.enabled = False

    def before(self, meta):
        if not self.enabled:
            return
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._cur = (e0, meta)

    def after(self):
        if not self.enabled or self._cur is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((self._cur[0], e1, self._cur[1]))
        self._cur = None


def algorithmic_bytes(meta, pair_count):
    K, Cin, Cout, M = meta["K"], meta["Cin"], meta["Cout"], meta["M"]
    P = pair_count
    return 4 * P * Cin + 4 * M * Cout + 8 * P + 4 * K * Cin * Cout, 2 * P * Cin * Cout


def build_model(query_num, device, decoder_dtype="fp32"):
    import segdino3d_amd as seg
    from segdino3d_amd.configs import scannet200_model_cfg
    torch.manual_seed(0)
    cfg = scannet200_model_cfg(query_num=query_num)
    cfg["decoder_cfg"]["compute_dtype"] = decoder_dtype
    model = seg.build_architecture(cfg).eval()
    # random-init weights of the architecture; give BN non-trivial running stats so nothing degenerates
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
    model.to(device)
    model.to_host = False
    return model


def cpu_baseline(model, scene_args, n_timed=4):
    """Oracle (CPU restatement) timed on the host cores with the SAME weights and scene shape."""
    from oracle import model_ref
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    # the oracle's many small ops oversubscribe a 128-core host; 32 threads is its best setting there
    threads = min(torch.get_num_threads(), int(os.environ.get("SD3D_CPU_THREADS", "32")))
    torch.set_num_threads(threads)
    times = []
    from segdino3d_amd.synth import make_scene
    for i in range(n_timed + 1):
        pts, tgt = make_scene(1000 + i, *scene_args)
        ef = tgt.extra_features
        t0 = time.perf_counter()
        model_ref.forward_eval(sd, pts, ef["points_2dfeats"], ef["super_point_masks"], ef["query2d_feats"],
                               ef["query2d_pos"], tgt.masks, query_num=model.query_num)
        dt = time.perf_counter() - t0
        if i > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": f"{n_timed} synthetic scenes of the benchmark shape after 1 warm-up, median {med:.2f} s/scene, "
                      f"torch CPU fp32 with {threads} threads (oracle/model_ref.forward_eval)"}


def end_to_end(model, device, args, n_scenes=96):
    """SURVEY 8(d) "report end-to-end separately": the same forward fed from packed scene FILES (disk / page cache -> pinned host ->
    H2D on a copy stream, segdino3d_amd.io_scene.ScenePrefetcher) with the post-processed PointData copied back to host numpy arrays
    (`model.to_host = True`: what evaluation/evaluate_3d.py:49-63 consumes).  Bounded: 4 packed scenes cycled, `n_scenes` forwards (~1 s:
    with 32 forwards - 0.35 s - the first file read and the drain of the last scenes were a fifth of the timed span and the rate read 85 - 93
    where 64 and more forwards read 108 - 114, `tools/e2e_parts.py`)."""
    import copy
    import shutil
    import tempfile
    from segdino3d_amd import io_scene
    from segdino3d_amd.dist_eval import PipelinedRunner
    from segdino3d_amd.synth import make_scene
    tmp = tempfile.mkdtemp(prefix="sd3d_e2e_")
    try:
        paths, nbytes = [], 0
        for j in range(4):
            pts, tgt = make_scene(500 + j, args.points, args.superpoints, args.query2d, layout=args.scene_layout)
            ef = tgt.extra_features
            path = os.path.join(tmp, f"s{j}.sd3d")
            nbytes = io_scene.pack_scene(path, dict(points=pts, super_points=ef["super_point_masks"], points_2dfeats=ef["points_2dfeats"],
                                                    query2d_feats=ef["query2d_feats"], query2d_pos=ef["query2d_pos"]))
            paths.append(path)
        runner = PipelinedRunner(model, args.streams, device)
        out_bytes = [0]

        def count(i, res):
            pd = res[0].pred_pts_seg
            out_bytes[0] += sum(a.nbytes for a in (pd.pts_instance_mask[0], pd.pts_instance_mask[1], pd.pts_semantic_mask[0],
                                                   pd.pts_semantic_mask[1], pd.instance_labels, pd.instance_scores))
        prev = model.to_host

        def timed(mode, n):
            model.to_host = mode
            out_bytes[0] = 0
            files = [paths[i % 4] for i in range(n)]
            t0 = time.perf_counter()
            # 159 MB per scene leave the page cache at 5 - 10 GB/s per reader thread: two readers cap the pipeline at ~80 scenes/s
            runner.run(io_scene.ScenePrefetcher(files, device, depth=8, readers=4), on_result=count, keep=False)
            torch.cuda.synchronize()
            return n / (time.perf_counter() - t0), out_bytes[0] // max(1, n)
        try:
            with torch.no_grad():
                model.to_host = "packed"
                # warm: files in the page cache, and as many pinned staging buffers in torch's host allocator as the timed runs will take
                runner.run(io_scene.ScenePrefetcher(paths * 4, device, depth=8, readers=4), keep=False)
                torch.cuda.synchronize()
                rate, nbytes_out = timed("packed", n_scenes)
                rate_bool, nbytes_bool = timed(True, max(8, n_scenes // 3))
        finally:
            model.to_host = prev
        return {"value": round(rate, 2), "unit": "scenes/s", "scenes": n_scenes, "input_bytes_per_scene": int(nbytes),
                "output_bytes_per_scene": int(nbytes_out),
                "unpacked_masks": {"value": round(rate_bool, 2), "output_bytes_per_scene": int(nbytes_bool),
                                   "what": "`to_host=True`: the [n, N] instance masks expanded to the bool array of the reference's evaluator "
                                           "(evaluator_3d.py:178) on the host, pageable"},
                "what": f"packed scene files (page cache) -> pinned host -> H2D (copy stream) -> forward, {args.streams} scenes in flight -> "
                        "post-processed masks (bit-packed on the device, `to_host=\"packed\"`) / labels / scores copied to pageable host arrays; "
                        "PCIe-inclusive, never `value`"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load_pmc_traffic(key):
    """HBM-side bytes per forward of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside this process) -
    ONLY if they were collected on the workload this run measures (`key`: scene shape, layout, query mode, scenes per forward).
    Returns (bytes per forward | None, source | reason)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            pmc = json.load(f)
        have = pmc.get("workload")
        if have is None:
            return None, f"profiles/{name} carries no workload key (collected before the traffic was keyed): not reported"
        if any(have.get(k) != v for k, v in key.items()):
            return None, f"profiles/{name} was collected on {have}, this run is {key}: not reported"
        return int(pmc.get("conv_bytes_per_forward", pmc.get("bytes_per_forward"))), pmc["source"]
    return None, "no PMC passes committed"


def pin_rank_to_cores(local_rank: int, local_world: int, threads_per_rank: int, share_gpu: bool = False):
    """One process per GPU drives `threads_per_rank` issuing threads; with 8 ranks on a node they must stay on the cores of
    the NUMA node their GPU hangs off (`/sys/bus/pci/devices/<bdf>/numa_node`) and off each other's cores: ranks whose GPUs
    share a node split that node's cores evenly (segdino3d_amd.dist_eval.cores_for_rank; an even slice of the allowed cores
    when the platform reports no node).  The GPUs' PCI addresses come from the KFD topology in sysfs
    (dist_eval.visible_gpu_bdfs): no HIP call touches the OTHER ranks' GPUs.  Returns (cores in the slice, NUMA node or -1)."""
    from segdino3d_amd.dist_eval import cores_for_rank, gpu_numa_node, visible_gpu_bdfs
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return 0, -1
    if local_world <= 1:
        return len(allowed), -1
    bdfs = visible_gpu_bdfs()
    nodes = []
    for r in range(local_world):
        g = 0 if share_gpu else r
        nodes.append(gpu_numa_node(bdfs[g]) if g < len(bdfs) else -1)
    mine = cores_for_rank(local_rank, nodes, allowed)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return len(allowed), nodes[local_rank]
    torch.set_num_threads(max(1, min(len(mine), threads_per_rank)))
    return len(mine), nodes[local_rank]


def launch_ranks(n_ranks: int, argv):
    """Parent of a self-launched multi-GPU run: starts one child per GPU and supervises all of them
    (segdino3d_amd.dist_eval.launch_ranks: rank 0's stdout - the JSON line - is relayed, the first rank that fails takes the
    others down at once).  The parent never touches the GPU: the devices are counted from the KFD topology in sysfs."""
    import socket
    from segdino3d_amd import dist_eval
    share = os.environ.get("SD3D_SHARE_GPU") == "1"
    n_dev = len(dist_eval.visible_gpu_bdfs())
    if n_dev < n_ranks and not share:
        print(f"bench.py: --gpus {n_ranks} but only {n_dev} GPU(s) are visible (SD3D_SHARE_GPU=1 + SD3D_DIST_BACKEND=gloo "
              f"rehearses the multi-rank flow on one GPU)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out0 = []
    rc = dist_eval.launch_ranks(dist_eval.rank_commands(os.path.abspath(__file__), argv, n_ranks, port), out0)
    sys.stdout.write(b"".join(out0).decode())
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--points", type=int, default=150_000)
    ap.add_argument("--superpoints", type=int, default=3000)
    ap.add_argument("--query2d", type=int, default=300)
    ap.add_argument("--query-num", type=int, default=200)
    ap.add_argument("--scene-pool", type=int, default=2, help="distinct synthetic scenes per rank (cycled)")
    ap.add_argument("--scene-layout", choices=("benchmark", "scan"), default="benchmark",
                    help="synth.make_scene layout: 'benchmark' = the scene of every reported number; 'scan' = mesh-like surface sampling")
    ap.add_argument("--streams", type=int, default=4,
                    help="scenes in flight per GPU (host threads x HIP streams, segdino3d_amd.dist_eval.PipelinedRunner)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SD3D_BENCH_BATCH", "5")),
                    help="scenes per forward: > 1 runs them as ONE block-diagonal sparse tensor (sparse.BatchSceneMaps + the batched decoder, every "
                         "scene's outputs bit-identical to its single-scene forward); each stream then keeps a whole batch in flight.  Default 5 "
                         "(round 6; 4 before): the driver's 20 steps on 4 streams are then ONE forward of 5 scenes per stream instead of forwards of 2 + 3 "
                         "(convolutions 7.10 -> 6.83 ms per scene, `value` 124.1 -> 126.3 in the same-box sweep profiles/r06_ab_driver_batch.txt); "
                         "`single_scene` is always one scene per forward")
    ap.add_argument("--decoder-dtype", choices=("fp32", "bf16"), default=os.environ.get("SD3D_DECODER_DTYPE", "fp32"),
                    help="bf16 = BASELINE configs[2]: bf16-MFMA attention contractions and projections, fp32 accumulation")
    ap.add_argument("--forward-sizes", default="", help="e.g. '2,3': every stream cuts its scenes into forwards of these sizes, cyclically, in "
                    "EVERY pipelined run of the process (counter passes on exactly the forward sizes another command line times: tools/pmc_run.sh)")
    ap.add_argument("--skip-single-scene", action="store_true", help="no one-scene-per-forward latency groups (counter passes: every forward "
                    "of the process then has one of --forward-sizes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the packed-files -> host-numpy-outputs measurement (`end_to_end`)")
    ap.add_argument("--preroll-seconds", type=float, default=2.0, help="untimed pipelined pre-roll before the timed K steps")
    ap.add_argument("--sustain-seconds", type=float, default=1.0,
                    help="after the timed K steps the same K-step list is repeated for this long: `sustained` (never `value`)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:            # started without a launcher: be the launcher
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:                                     # never print a line whose n_gpus differs from --gpus
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SD3D_DIST_BACKEND=gloo + SD3D_SHARE_GPU=1: rehearsal of the multi-rank flow on ONE GPU (ranks share device 0,
    # collectives on CPU tensors); the real runs use "nccl" (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("SD3D_DIST_BACKEND", "nccl")
    share_gpu = os.environ.get("SD3D_SHARE_GPU") == "1"
    dev_index = 0 if share_gpu else local_rank
    cores_per_rank, numa_node = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), args.streams, share_gpu)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_device = device if backend == "nccl" else torch.device("cpu")
    dist = None
    # a launcher's environment (RANK + MASTER_ADDR / MASTER_PORT) means "be a rank" also at world size 1: the process group, the barriers, the
    # max-over-ranks all-reduce and the closing all-gather then run on the chosen backend exactly as they do at N > 1
    # (tests/test_gpu_rccl.py: the RCCL branch on the one GPU a box has).  A plain `python bench.py` (no RANK) stays group-free.
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ and "MASTER_PORT" in os.environ
    if world > 1 or launched:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    from segdino3d_amd import ops
    from segdino3d_amd.synth import make_scene

    model = build_model(args.query_num, device, args.decoder_dtype)
    scene_args = (args.points, args.superpoints, args.query2d)
    pool = []
    for j in range(max(1, args.scene_pool)):
        pts, tgt = make_scene(rank * 100 + j, *scene_args, layout=args.scene_layout)
        pool.append((pts.to(device), tgt.to(device)))

    def step(i):
        pts, tgt = pool[i % len(pool)]
        return model([pts], [tgt])

    from segdino3d_amd.dist_eval import PipelinedRunner
    runner = PipelinedRunner(model, args.streams, device, batch=args.batch)
    if args.forward_sizes:
        runner.forward_sizes = [int(v) for v in args.forward_sizes.split(",") if v.strip()]
    # every step gets its own target object (the forward attaches its outputs to it)
    import copy
    def scene_list(n):
        return [(pool[i % len(pool)][0], copy.copy(pool[i % len(pool)][1])) for i in range(n)]

    with torch.no_grad():
        for i in range(0 if args.skip_single_scene else args.warmup):
            step(i)
        if args.skip_single_scene:
            runner.run(scene_list(max(1, args.warmup)))
        runner.run(scene_list(max(args.streams, 2) * args.batch))   # warm the worker streams' allocator pools
        torch.cuda.synchronize()
        # single-scene latency (one stream, back to back): median over four groups of five forwards (one group of five read
        # 12.5 and 13.5 ms in two runs of the same build on the same box: a host hiccup in a 60 ms window is a 8 % error)
        n_lat, groups = min(5, args.steps), []
        for _ in range(0 if args.skip_single_scene else 4):
            t0 = time.perf_counter()
            for i in range(n_lat):
                step(i)
            torch.cuda.synchronize()
            groups.append(1e3 * (time.perf_counter() - t0) / n_lat)
        latency_ms = (sorted(groups)[1:3][0] * 0.5 + sorted(groups)[1:3][1] * 0.5) if groups else float("nan")
        # pipelined pre-roll (untimed): the W warm-up steps above ran on one stream; the timed region runs `streams` host
        # threads, whose allocator pools, code objects and - on a freshly booted node - host clocks need a second of the
        # real workload to settle (a cold node measured 85 -> 93 -> 98 scenes/s over three back-to-back processes without it)
        t_pre, n_pre = time.perf_counter(), 0
        while time.perf_counter() - t_pre < args.preroll_seconds:
            runner.run(scene_list(2 * args.streams * args.batch))
            n_pre += 2 * args.streams * args.batch
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        work = scene_list(args.steps)
        plan = runner.plan_batches(args.steps)                    # [[scene ids of a forward, ...] per stream]: what the timed region runs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(work)                                          # EXACTLY K steps, args.streams forwards in flight
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        # sustained rate: the same K-step list repeated until >= 1 s has been timed (the K = 20 region of the driver is 0.2 s, a
        # sixth of it pipeline ramp and drain); reported next to `value`, never instead of it
        sus_n, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.sustain_seconds:
            runner.run(scene_list(args.steps))
            sus_n += args.steps
        torch.cuda.synchronize()
        sus_dt = time.perf_counter() - t1
        # one scene per forward, for the per-scene voxel counts of the line
        import segdino3d_amd as seg
        with seg.capture() as cap1:
            if args.skip_single_scene:                            # (still no one-scene forward in the process)
                ids0 = plan[0][0]
                model([work[i][0] for i in ids0], [copy.copy(work[i][1]) for i in ids0])
            else:
                step(0)
        # Instrumented replay of the same K steps for the roofline of the dominant kernel: a HIP-event
        # pair around every gather_gemm launch costs ~2 x 237 event records per step (+15-20 % wall),
        # so it is kept out of the region that produces `value`.  It runs the forwards of `plan` - the same scenes grouped
        # into the same forwards as the timed region - one after the other on one stream.
        timer = GemmTimer()
        ops.GG_HOOK = timer
        timer.enabled = True
        n_fwd = 0
        for st_plan in plan:
            for ids in st_plan:
                model([work[i][0] for i in ids], [copy.copy(work[i][1]) for i in ids])
                n_fwd += 1
        torch.cuda.synchronize()
        timer.enabled = False
        ops.GG_HOOK = None

    # max over ranks
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_scenes = world * args.steps
    value = total_scenes / dt

    # ---- roofline of the dominant kernel (rank 0's launches) -------------------------------------------
    gemm_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in timer.records)
    n_launch = len(timer.records)
    pair_cache = {}
    tot_bytes = tot_flops = 0
    conv_ms = conv_flops = conv_bytes = conv_n = 0      # the sparse convolutions alone (sd3d_pair_conv: pass 1 + pass 2)
    for e0_, e1_, meta in timer.records:
        nbr = meta["nbr"]
        if meta.get("pairs") is not None:              # pair-major convolution: count the real list entries (the REAL tiles: an evaluation
            pl = meta["pairs"]                          # table's capacity behind them is unwritten memory, segdino3d_amd/sparse.py LEAN_LISTS)
            key = (pl.in_idx.data_ptr(), pl.p_cap)
            if key not in pair_cache:
                n_real = int(pl.tile_k[pl.p_cap // 128].item())
                pair_cache[key] = int((pl.in_idx[: n_real * 128] >= 0).sum().item())
            nbr = pl.in_idx.new_empty(0)
            nbr_key = key
        else:
            nbr_key = None
        if nbr is None:
            P = meta["M"]
        elif nbr_key is not None:
            P = pair_cache[nbr_key]
        else:
            key = (nbr.data_ptr(), tuple(nbr.shape))
            if key not in pair_cache:
                pair_cache[key] = int((nbr >= 0).sum().item())
            P = pair_cache[key]
        b, f = algorithmic_bytes(meta, P)
        tot_bytes += b
        tot_flops += f
        if meta.get("pairs") is not None:
            conv_ms += e0_.elapsed_time(e1_); conv_flops += f; conv_bytes += b; conv_n += 1
    sec = gemm_ms * 1e-3
    fam_gbs = tot_bytes / sec / 1e9 if sec > 0 else 0.0
    fam_tf = tot_flops / sec / 1e12 if sec > 0 else 0.0
    steps = max(1, args.steps)
    conv_sec = conv_ms * 1e-3
    conv_tf = conv_flops / conv_sec / 1e12 if conv_sec > 0 else 0.0
    conv_gbs = conv_bytes / conv_sec / 1e9 if conv_sec > 0 else 0.0
    n_fwd = max(1, n_fwd)                                       # forwards of the instrumented replay = forwards of the timed region
    conv_per_step = max(1, conv_n // n_fwd)
    # HBM-side bytes of the dominant kernel per forward: FETCH_SIZE x 2 (the guide's gfx950 correction) + WRITE_SIZE of pair_gemm_* +
    # pair_reduce_* from the committed PMC passes, reported only when they were collected on THIS workload.  They exceed the
    # algorithmic bytes by design: the partial products are written by pass 1 and re-read by pass 2.
    # keyed on the forward sizes the timed region REALLY ran (the driver's 20 steps on 4 streams x batches of <= 4 are forwards of 2 and 3
    # scenes): the counter passes replay those sizes on one stream (`tools/pmc_run.sh ... --forward-sizes 2,3 --skip-single-scene`)
    workload_key = {"points": args.points, "superpoints": args.superpoints, "queries_2d": args.query2d, "scene_layout": args.scene_layout,
                    "forward_sizes": sorted({len(f) for st_plan in plan for f in st_plan})}
    traffic_fwd, traffic_src = load_pmc_traffic(workload_key)
    suspect = conv_tf > FP32_MFMA_PEAK_TFLOPS or fam_tf > FP32_MFMA_PEAK_TFLOPS
    if suspect:   # more flops than the matrix pipe can issue: the accounting, not the GPU (e.g. list entries counted beyond the real tiles)
        print(f"bench.py: roofline accounting is off: {conv_tf:.1f} / {fam_tf:.1f} TFLOP/s exceed the fp32 matrix peak", file=sys.stderr)
    roofline = {"bound": "mfma", **({"accounting_suspect": True} if suspect else {}),
                "kernel": "sd3d_pair_conv_ex = pair_gemm_* (pass 1, fp32 MFMA over the offset-major rulebook) + pair_reduce_rl_kernel (pass 2 over per-row lists; "
                          "none for the transposed convolutions): the 55 sparse convolutions of Res16UNet34C",
                "achieved": round(conv_tf, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(conv_tf / FP32_MFMA_PEAK_TFLOPS, 4),
                "launches_per_forward": conv_per_step, "ms_per_forward": round(conv_ms / steps, 3),
                "algorithmic_flops_per_forward": conv_flops // steps, "algorithmic_bytes_per_forward": conv_bytes // steps,
                "traffic": traffic_fwd, "traffic_unit": "HBM-side bytes per forward of the 55 convolutions (PMC: FETCH_SIZE x 2 + WRITE_SIZE)",
                "traffic_over_algorithmic": round(traffic_fwd / max(1, conv_bytes // steps), 2) if traffic_fwd else None,
                "traffic_source": traffic_src,
                "hbm_achieved_gbs": round(conv_gbs, 1), "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": round(conv_gbs / HBM_PEAK_GBS, 4),
                # the same with the MEASURED HBM-side bytes (partial products written by pass 1 and re-read by pass 2 included)
                "hbm_traffic_gbs": round(traffic_fwd / (conv_ms / steps) / 1e6, 1) if (traffic_fwd and conv_ms > 0) else None,
                "hbm_traffic_frac": round(traffic_fwd / (conv_ms / steps) / 1e6 / HBM_PEAK_GBS, 4) if (traffic_fwd and conv_ms > 0) else None,
                "share_of_single_stream_forward": round(conv_ms / steps / latency_ms, 3) if (args.batch == 1 and groups) else None,
                "measured": "HIP events around every launch on the launching stream, single-stream instrumented replay of the timed steps"
                            + (" in the forwards of `config.forward_sizes` (`*_per_forward` figures are per SCENE, `launches_per_forward` per forward call)"
                               if args.batch > 1 else ""),
                # every sparse convolution AND every Linear of the decoder / heads (the ~155 extra launches are 10 us each for 26 MFLOP)
                "gemm_family": {"launches_per_forward": n_launch // n_fwd, "ms_per_forward": round(gemm_ms / steps, 3),
                                "ms_per_forward_note": "single-stream SUM of HIP-event pairs around every launch (each pair reads >= launch + drain of a "
                                                       "10 us kernel): not a share of `ms_per_step`, and it may exceed it",
                                "achieved": round(fam_tf, 2), "unit": "TFLOP/s", "frac": round(fam_tf / FP32_MFMA_PEAK_TFLOPS, 4),
                                "hbm_achieved_gbs": round(fam_gbs, 1), "algorithmic_bytes_per_forward": tot_bytes // steps,
                                "algorithmic_flops_per_forward": tot_flops // steps}}

    # ---- closing all-gather of per-scene records over RCCL/xGMI -----------------------
    maps = cap1.maps[-1]                                      # ONE scene's maps: per-scene voxel counts
    fwd_sizes = [[len(f) for f in st] for st in plan]
    rec = torch.tensor([float(rank), float(args.points), float(maps.n_vox[0]), 1e3 * dt / args.steps],
                       dtype=torch.float64, device=comm_device)
    if dist is not None:
        gathered = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(gathered, rec)
        records = torch.stack(gathered).cpu().tolist()
    else:
        records = [rec.cpu().tolist()]

    cpu = e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        e2e = end_to_end(model, device, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, scene_args)

    if rank == 0:
        bf16_dec = getattr(model.decoder, "compute_dtype", "fp32") == "bf16"      # SD3D_DECODER_DTYPE=bf16: BASELINE configs[2]
        out = {
            "metric": "scenes/sec forward (ScanNet200 ~150k pts, " + ("200 queries" if args.query_num == 200 else
                                                                       ("one query per superpoint" if args.query_num < 0 else f"{args.query_num} queries")) + ")",
            "value": round(value, 3), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 backbone + bf16 decoder contractions" if bf16_dec else "f32", "data": "synthetic",
            "single_scene": None if not groups else {
                "scenes_per_s": round(1e3 / latency_ms, 2), "latency_ms": round(latency_ms, 3),
                "latency_ms_groups_of_5": [round(g, 3) for g in groups],
                "note": "ONE scene in flight per GPU (SURVEY 8(d) batch = 1), same forward, same process"},
            "config": {"workload": ("configs[2]" if bf16_dec else "configs[1]") + ": ScanNet-val-like scenes; a STEP is one scene; `value` times the K scenes as "
                                   f"{sum(len(st) for st in fwd_sizes)} evaluation forwards of {'/'.join(str(v) for v in workload_key['forward_sizes'])} scenes each "
                                   f"(`forward_sizes`), {args.streams} forward(s) in flight per GPU (<= {sum(max(st) if st else 0 for st in fwd_sizes)} scenes in flight); "
                                   "`single_scene` = one forward of one scene in flight; "
                                   "fp32 sparse backbone (Res16UNet34C) + " + ("bf16-MFMA" if bf16_dec else "fp32") +
                                   " decoder + post-processing, device-resident in/out",
                       "points": args.points, "superpoints": args.superpoints, "queries_2d": args.query2d, "scene_layout": args.scene_layout,
                       "query_num": args.query_num, "voxels_per_level": maps.n_vox, "parallelism": f"scene-sharded x{world}",
                       "streams": args.streams, "max_scenes_per_forward": args.batch,
                       "forward_sizes": fwd_sizes,             # per stream: scenes of each forward of the timed region (PipelinedRunner.plan_batches)
                       "scenes_in_flight_per_gpu": sum(max(st) if st else 0 for st in fwd_sizes),
                       "scenes_per_forward": round(args.steps / max(1, sum(len(st) for st in fwd_sizes)), 2),
                       "single_stream_latency_ms": round(latency_ms, 3) if groups else None,
                       "untimed_preroll_scenes": n_pre, "host_cores_per_rank": cores_per_rank, "gpu_numa_node": numa_node},
            "sustained": {"scenes_per_s": round(world * sus_n / sus_dt, 3) if sus_n else None, "timed_scenes": sus_n, "seconds": round(sus_dt, 3),
                          "note": "the same K-step list repeated back to back after the K timed steps (rank 0's clock); not `value`"},
            "roofline": roofline, "cpu_baseline": cpu, "end_to_end": e2e,
            "per_rank_records": records,
            "process_group": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                        "collectives": "barrier x2, all_reduce(MAX) of the timed span, all_gather of the per-rank records"},
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
