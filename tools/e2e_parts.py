"""Which side bounds `end_to_end`?  (1) the input side alone (packed files -> pinned -> H2D, no forward), (2) the forward with device-resident
inputs and host outputs (`to_host="packed"`), (3) both - same process, 4 streams, one scene per forward as bench.py's end_to_end runs it.
usage: python tools/e2e_parts.py [readers ...]"""
import copy
import os
import shutil
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench
from segdino3d_amd import io_scene
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.synth import make_scene

d = torch.device("cuda:0")
model = bench.build_model(200, d)
tmp = tempfile.mkdtemp(prefix="sd3d_e2e_")
try:
    paths = []
    for j in range(4):
        pts, tgt = make_scene(500 + j, 150000, 3000, 300)
        ef = tgt.extra_features
        p = os.path.join(tmp, f"s{j}.sd3d")
        io_scene.pack_scene(p, dict(points=pts, super_points=ef["super_point_masks"], points_2dfeats=ef["points_2dfeats"],
                                    query2d_feats=ef["query2d_feats"], query2d_pos=ef["query2d_pos"]))
        paths.append(p)
    R = 64
    files = [paths[i % 4] for i in range(R)]
    runner = PipelinedRunner(model, 4, d)
    model.to_host = "packed"
    with torch.no_grad():
        warm = [(p, copy.copy(t)) for p, t in io_scene.ScenePrefetcher(paths, d, depth=2)]
        runner.run(warm + warm, keep=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run([(warm[i % 4][0], copy.copy(warm[i % 4][1])) for i in range(R)], keep=False)
        torch.cuda.synchronize()
        print(f"forward, device-resident inputs, packed host outputs: {R / (time.perf_counter() - t0):.1f} scenes/s")
        for readers in [int(a) for a in sys.argv[1:]] or [2, 4, 6]:
            for rep in range(2):
                t0 = time.perf_counter()
                n = 0
                for pts_, tgt_ in io_scene.ScenePrefetcher(files, d, depth=8, readers=readers):
                    n += 1
                torch.cuda.synchronize()
                t_in = R / (time.perf_counter() - t0)
                t0 = time.perf_counter()
                runner.run(io_scene.ScenePrefetcher(files, d, depth=8, readers=readers), keep=False)
                torch.cuda.synchronize()
                print(f"readers {readers}: input side alone {t_in:.1f} scenes/s, end to end {R / (time.perf_counter() - t0):.1f} scenes/s")
finally:
    shutil.rmtree(tmp, ignore_errors=True)
