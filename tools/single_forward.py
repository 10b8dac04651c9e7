"""Single-scene forwards, one in flight (the `single_scene` operating point), for a kernel timeline:
   rocprofv3 --kernel-trace -d /tmp/tl -o r -- python3 tools/single_forward.py [n]      then tools/timeline_forward.py <db>"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

import bench
from segdino3d_amd.synth import make_scene

d = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
model = bench.build_model(int(os.environ.get("QUERY_NUM", "200")), d)
pool = [tuple(t.to(d) for t in make_scene(j, 150000, 3000, 300)) for j in range(2)]
with torch.no_grad():
    for i in range(4):
        model([pool[i % 2][0]], [pool[i % 2][1]])
    torch.cuda.synchronize()
    time.sleep(0.05)                                           # a gap the timeline tool finds: the forwards after it are the sample
    t0 = time.perf_counter()
    for i in range(n):
        model([pool[i % 2][0]], [pool[i % 2][1]])
    torch.cuda.synchronize()
    print(f"{1e3 * (time.perf_counter() - t0) / n:.3f} ms per forward, one scene in flight")
