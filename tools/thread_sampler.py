"""Where are the worker threads of the pipelined runner?  Samples their Python stacks every 0.5 ms."""
import os, sys, time, copy, threading, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd.dist_eval import PipelinedRunner
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pts, tgt = make_scene(0, 150000, 3000, 300)
pts = pts.to(d); tgt = tgt.to(d)
runner = PipelinedRunner(model, NS, d)
scenes = lambda n: [(pts, copy.copy(tgt)) for _ in range(n)]
runner.run(scenes(6)); torch.cuda.synchronize()
hist = collections.Counter(); stop = False; me = threading.get_ident(); nsamp = 0
def classify(frame):
    names = []
    f = frame
    while f is not None:
        names.append((os.path.basename(f.f_code.co_filename), f.f_code.co_name))
        f = f.f_back
    top = names[0]
    for fn, nm in names:
        if nm in ("wait_event",): return "wait_event (poll GPU)"
    for fn, nm in names:
        if nm in ("_select", "nonzero"): return "post: data-dependent selects"
        if fn == "decoder.py": return "decoder issue"
        if nm == "predict_by_feat" or nm == "_instances_common": return "post issue"
        if nm == "run" and fn == "plan.py": return "unet plan (C call, GIL released)"
        if fn == "sparse.py": return "voxelise / tables issue"
        if fn in ("backbone_mink.py",): return "backbone other"
    return "other:" + top[1]
def sampler():
    global nsamp
    while not stop:
        for tid, fr in sys._current_frames().items():
            if tid == me or tid == threading.get_ident(): continue
            hist[classify(fr)] += 1
        nsamp += 1
        time.sleep(0.0005)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter()
R = 45
runner.run(scenes(R)); torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"streams {NS}: {1e3*dt/R:.2f} ms/scene; {nsamp} samples")
tot = sum(hist.values())
for k, v in hist.most_common(12):
    print(f"  {100*v/tot:5.1f}%  {k}")
