"""Upper bound of what a HIP graph could give the U-Net of ONE scene: the ~170 launches of `LayerPlan.run` captured once and replayed, against the
same launches issued through the stream (same tables, same buffers).  usage: python tools/unet_graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench
from segdino3d_amd import ops, sparse
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene

d = torch.device("cuda:0")
model = bench.build_model(200, d)
bb = model.backbone
pts, tgt = make_scene(0, 150000, 3000, 300)
pts, tgt = pts.to(d), tgt.to(d)
sparse.FORK_JOIN = False
with torch.no_grad():
    model([pts], [tgt])                                         # builds the plan, warms everything
    maps, vf, _, _, _, _ = bb._scene_inputs(pts.float().contiguous(), tgt)
    k1 = bb.conv1_kernel_size
    maps.prepare(same=[(0, k1)] + [(l, 3) for l in range(5)], strides=[0, 1, 2, 3], chained=True)
    plan = bb._plan
    torch.cuda.synchronize()

    def direct(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = plan.run(maps, vf)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    direct(3)
    t_direct = direct(10)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), ops.use_stream(s):
        plan.run(maps, vf)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out_g = plan.run(maps, vf)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    t_graph = e0.elapsed_time(e1) / 10
    t_direct2 = direct(10)
print(f"U-Net of one scene, ms per run: stream launches {t_direct:.3f} / {t_direct2:.3f}, graph replay {t_graph:.3f}")
