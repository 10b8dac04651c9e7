"""Batched evaluation forward (sparse.BatchSceneMaps) against single-scene forwards: bit-identity of every output and
phase times (HIP events) of one batch of B scenes next to B sequential scenes, then scenes/s of the pipelined runner for a
few (streams, batch) settings.   python tools/batch_quick.py [B]"""
import copy, os, sys, time, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import segdino3d_amd as seg
from segdino3d_amd import plan, sparse
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.synth import make_scene

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
qn = int(os.environ.get("QN", "200"))
d = torch.device("cuda:0")
model = bench.build_model(qn, d)
pool = [tuple(t.to(d) for t in make_scene(j, 150000 - 3000 * j, 3000 - 50 * j, 300)) for j in range(B)]
marks = []


def wrap(obj, name, label):
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


def fresh():
    return [p for p, _ in pool], [copy.copy(t) for _, t in pool]


def fields(pd):
    out = []
    for k, v in sorted(pd.items()):
        vs = v if isinstance(v, (list, tuple)) else [v]
        out += [(k, x) for x in vs if isinstance(x, torch.Tensor)]
    return out


with torch.no_grad():
    # ---- bit-identity ----
    singles, caps = [], []
    for p, t in zip(*fresh()):
        with seg.capture() as cap:
            singles.append(model([p], [t])[0])
        caps.append(cap)
    pts, tgts = fresh()
    with seg.capture() as capb:
        batched = model(pts, tgts)
    torch.cuda.synchronize()
    ok = True
    for i in range(B):
        same_sp = torch.equal(caps[i].sp_feats[0], capb.sp_feats[i])
        same_logits = torch.equal(caps[i].outputs["masks"][0], capb.outputs["masks"][i])
        fa, fb = fields(singles[i].pred_pts_seg), fields(batched[i].pred_pts_seg)
        same_out = len(fa) == len(fb) and all(ka == kb and a.shape == b.shape and torch.equal(a, b) for (ka, a), (kb, b) in zip(fa, fb))
        print(f"scene {i}: superpoint features equal {same_sp}, mask logits equal {same_logits}, post-processed outputs equal {same_out}"
              f" ({len(fa)} tensors, {singles[i].pred_pts_seg.instance_scores.shape[0]} instances)")
        if not same_sp:
            diff = (caps[i].sp_feats[0] - capb.sp_feats[i]).abs()
            print(f"   max |diff| {diff.max().item():.3e}, rows differing {(diff > 0).any(dim=1).sum().item()} of {diff.shape[0]}")
        ok &= same_sp and same_logits and same_out
    print("BIT-IDENTICAL" if ok else "MISMATCH")

    # ---- phase times ----
    wrap(sparse.SceneMaps, "__init__", "1 voxelise + levels (single)")
    wrap(sparse.BatchSceneMaps, "__init__", "1 voxelise + levels (batch)")
    wrap(sparse.SceneMaps, "prepare", "2 neighbour tables + pair lists")
    wrap(plan.LayerPlan, "run", "3 U-Net (sd3d_run_layers)")
    wrap(model.decoder, "forward", "5 decoder")
    wrap(model, "predict_by_feat", "6 post-processing")
    wrap(model, "forward", "0 whole forward")

    def report(tag, n_scenes, wall):
        torch.cuda.synchronize()
        agg = {}
        for label, e0, e1 in marks:
            agg.setdefault(label, []).append(e0.elapsed_time(e1))
        print(f"== {tag}: {n_scenes / wall:.1f} scenes/s, {1e3 * wall / n_scenes:.2f} ms per scene")
        for label in sorted(agg):
            v = agg[label]
            print(f"   {label:36s} {sum(v) / len(v):8.2f} ms per call, {sum(v) / n_scenes:7.2f} ms per scene (n={len(v)})")
        marks.clear()

    for _ in range(2):
        model(*fresh())
    torch.cuda.synchronize(); marks.clear()
    R = 6
    t0 = time.perf_counter()
    for _ in range(R):
        for p, t in zip(*fresh()):
            model([p], [t])
    torch.cuda.synchronize()
    report("sequential single scenes, one stream", R * B, time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(R):
        model(*fresh())
    torch.cuda.synchronize()
    report(f"batches of {B}, one stream", R * B, time.perf_counter() - t0)

    def scene_list(n):
        return [(pool[i % B][0], copy.copy(pool[i % B][1])) for i in range(n)]

    for streams, batch in ((4, 1), (1, B), (2, B), (3, B), (2, 2), (4, 2)):
        r = PipelinedRunner(model, streams, d, batch=batch)
        r.run(scene_list(2 * streams * batch))
        torch.cuda.synchronize()
        n = 24 * max(1, (streams * batch) // 4) * 4
        t0 = time.perf_counter()
        r.run(scene_list(n))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"streams {streams} x batch {batch}: {n / dt:.1f} scenes/s ({1e3 * dt / n:.2f} ms per scene)")
