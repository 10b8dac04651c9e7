"""Same-process A/B of module switches at the `single_scene` operating point (one scene in flight, back to back): the settings
alternate every 10 forwards for `rounds` rounds, the report is the median / min of the group means - box-to-box and process-to-process
noise (0.3 - 0.5 ms between two bench.py runs of one build) cancels.
usage: python tools/ab_single.py [rounds] [query_num]   (switches: sparse.FORK_JOIN, sparse.OPTIMISTIC_SORT, decoder.FUSED_NARROW)"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench
from segdino3d_amd import decoder, sparse
from segdino3d_amd.synth import make_scene

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
qn = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = torch.device("cuda:0")
model = bench.build_model(qn, d)
pool = [tuple(t.to(d) for t in make_scene(j, 150000, 3000, 300)) for j in range(2)]
SETTINGS = {
    "base (no fork, op-by-op decoder)": dict(fork=False, narrow=False),
    "fork / join": dict(fork=True, narrow=False),
    "narrow row chain": dict(fork=False, narrow=True),
    "fork / join + narrow row chain": dict(fork=True, narrow=True),
    "fork / join, full radix sorts": dict(fork=True, narrow=False, optimistic=False),
    "fork / join, one unique per level": dict(fork=True, narrow=False, levels=False),
    "fork / join, voxelisation by separate calls": dict(fork=True, narrow=False, one_call=False),
}
only = os.environ.get("AB_ONLY")
if only:
    SETTINGS = {k: v for k, v in SETTINGS.items() if any(s in k for s in only.split(","))}


def apply(s):
    sparse.FORK_JOIN = s["fork"]
    sparse.OPTIMISTIC_SORT = s.get("optimistic", True)
    sparse.LEVELS_AT_ONCE = s.get("levels", True)
    sparse.VOXELISE_ONE_CALL = s.get("one_call", True)
    decoder.FUSED_NARROW = s["narrow"]


# prologue = GPU time from the forward's first launch to the point where the U-Net's first layer may start (an event on the scene's stream
# right before `LayerPlan.run` enqueues it): what the voxelisation / map / list work in front of the convolutions costs the scene
from segdino3d_amd import plan as _plan
_marks = []
_run = _plan.LayerPlan.run


def _run_marked(self, maps, x):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    _marks.append(e)
    return _run(self, maps, x)


_plan.LayerPlan.run = _run_marked
prologue = {}


def group(n=10, key=None):
    torch.cuda.synchronize()
    _marks.clear()
    begins = []
    t0 = time.perf_counter()
    for i in range(n):
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        begins.append(b)
        model([pool[i % 2][0]], [pool[i % 2][1]])
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / n
    if key is not None and len(_marks) == n:
        prologue.setdefault(key, []).append(statistics.median(b.elapsed_time(e) for b, e in zip(begins, _marks)))
    return dt


res = {k: [] for k in SETTINGS}
with torch.no_grad():
    for s in SETTINGS.values():
        apply(s)
        group(4)
    for r in range(rounds):
        for k, s in SETTINGS.items():
            apply(s)
            res[k].append(group(key=k))
for k, v in res.items():
    pro = f"   prologue (forward start -> first convolution may start) median {statistics.median(prologue[k]):6.3f} ms" if k in prologue else ""
    print(f"{k:40s} median {statistics.median(v):7.3f} ms   min {min(v):7.3f}   max {max(v):7.3f}   (groups of 10 forwards, {len(v)} rounds){pro}")
if os.environ.get("AB_HOST") == "1":                          # host time of the phases that precede the first convolution
    from segdino3d_amd import plan
    acc = {}

    def wrap(obj, name, label):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc.setdefault(label, []).append(1e6 * (time.perf_counter() - t))
        setattr(obj, name, g)
    wrap(sparse.SceneMaps, "__init__", "SceneMaps.__init__ (voxelise, levels, sync 1)")
    wrap(sparse.SceneMaps, "prepare", "SceneMaps.prepare (tables + lists)")
    wrap(plan.LayerPlan, "run", "LayerPlan.run (enqueue the U-Net)")
    wrap(model.decoder, "forward", "decoder.forward")
    wrap(model, "predict_by_feat", "predict_by_feat")
    wrap(model, "forward", "whole forward (host returns)")
    with torch.no_grad():
        for k, s in SETTINGS.items():
            apply(s)
            acc.clear()
            group(10)
            print(f"host us per forward, {k}:", {lab: round(statistics.median(v)) for lab, v in acc.items()})
if os.environ.get("AB_PHASE") == "1":                         # device time of the phases, HIP events on the scene's stream
    from segdino3d_amd import plan
    marks = []

    def wrap_ev(obj, name, label):
        f = getattr(obj, name)

        def g(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                return f(*a, **k)
            finally:
                e1.record()
                marks.append((label, e0, e1))
        setattr(obj, name, g)
    wrap_ev(sparse.SceneMaps, "__init__", "1 voxelise + levels (to sync 1)")
    wrap_ev(sparse.SceneMaps, "prepare", "2 tables + lists, this stream")
    wrap_ev(plan.LayerPlan, "run", "3 U-Net")
    wrap_ev(model.decoder, "forward", "5 decoder")
    wrap_ev(model, "predict_by_feat", "6 post-processing")
    wrap_ev(model, "forward", "0 whole forward")
    with torch.no_grad():
        for k, s in SETTINGS.items():
            apply(s)
            group(3)
            marks.clear()
            group(10)
            agg = {}
            for label, e0, e1 in marks:
                agg.setdefault(label, []).append(e0.elapsed_time(e1))
            print(f"device ms per phase, {k}: " + ", ".join(f"{lab}: {statistics.median(v):.3f}" for lab, v in sorted(agg.items())))
