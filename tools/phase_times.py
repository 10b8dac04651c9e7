"""Wall time per phase of one eval forward (synchronised between phases) - where the milliseconds go."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from segdino3d_amd import ops
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene

d = torch.device("cuda:0")
model = bench.build_model(200, d)
pts, tgt = make_scene(0, 150000, 3000, 300)
pts, tgt = pts.to(d), tgt.to(d)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for _ in range(3):
        model([pts], [tgt])
    acc = {}
    for it in range(5):
        t0 = sync()
        sr = model.get_extra_instance_data([pts], [tgt], True, True)
        t1 = sync()
        bb = model.backbone
        ef = tgt["extra_features"]
        maps = SceneMaps(pts, bb.voxel_size, 5, superpoints=ef["super_point_masks"])
        t2 = sync()
        vf = maps.voxel_features(pts, ef["points_2dfeats"], 0, 288)
        t3 = sync()
        maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
        t4 = sync()
        x = bb.forward_sparse(maps, vf)
        t5 = sync()
        f, p = maps.pool(x, 96)
        t6 = sync()
        q, qp, _ = model._select_queries([f], [p], [tgt])
        t7 = sync()
        out = model.forward_decoder([f], [p], [p], q, qp, [tgt], sr)
        t8 = sync()
        pred = model.predict_by_feat([pts], out, ef["super_point_masks"])
        t9 = sync()
        names = ["extra_instance_data", "voxelise+levels", "voxel_mean", "kernel maps", "sparse convs", "pool", "select_queries", "decoder", "post-process"]
        ts = [t0, t1, t2, t3, t4, t5, t6, t7, t8, t9]
        for n, a, b in zip(names, ts[:-1], ts[1:]):
            acc[n] = acc.get(n, 0.0) + (b - a) * 1e3 / 5
    tot = 0
    for n, v in acc.items():
        print(f"{n:22s} {v:8.3f} ms"); tot += v
    print(f"{'sum':22s} {tot:8.3f} ms")
    t0 = sync()
    for _ in range(5):
        model([pts], [tgt])
    print(f"{'unsynchronised forward':22s} {(sync() - t0) * 1e3 / 5:8.3f} ms")
    # sparse-conv phase: wall vs stream time vs sum of per-launch event pairs
    timer = bench.GemmTimer(); ops.GG_HOOK = timer
    maps = SceneMaps(pts, bb.voxel_size, 5, superpoints=ef["super_point_masks"])
    vf = maps.voxel_features(pts, ef["points_2dfeats"], 0, 288)
    maps.prepare(same=[(0, 5)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3])
    for instrument in (False, True):
        timer.enabled = instrument; timer.records.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = sync(); e0.record()
        x = bb.forward_sparse(maps, vf)
        t_host = time.perf_counter(); e1.record(); t1 = sync()
        per = sum(a.elapsed_time(b) for a, b, _ in timer.records)
        print(f"convs instrument={instrument}: wall {1e3 * (t1 - t0):.2f} ms, host enqueue {1e3 * (t_host - t0):.2f} ms, "
              f"stream {e0.elapsed_time(e1):.2f} ms, sum of {len(timer.records)} launch events {per:.2f} ms")
    ops.GG_HOOK = None
