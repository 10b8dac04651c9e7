"""Per-launch table of the gather-GEMM family over one ScanNet200 forward (HIP events, single stream):
shape, rulebook density, time, active TFLOP/s and algorithmic GB/s, grouped by identical shape."""
import os, sys, collections, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from segdino3d_amd import ops
from segdino3d_amd.synth import make_scene
d = torch.device("cuda:0")
model = bench.build_model(200, d)
pts, tgt = make_scene(0, 150000, 3000, 300)
pts = pts.to(d); tgt = tgt.to(d)
timer = bench.GemmTimer()
ops.GG_HOOK = timer
with torch.no_grad():
    for _ in range(3):
        model([pts], [tgt])
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    REP = 5
    for _ in range(REP):
        timer.records.clear(); timer.enabled = True
        model([pts], [tgt])
        torch.cuda.synchronize(); timer.enabled = False
        for i, (e0, e1, meta) in enumerate(timer.records):
            nbr = meta["nbr"] if meta.get("pairs") is None else meta["pairs"].in_idx
            pairs = int((nbr >= 0).sum()) if nbr is not None else meta["M"]
            key = (i, meta["K"], meta["Cin"], meta["Cout"], meta["M"], pairs, meta.get("pairs") is not None)
            agg.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
rows = []
for (i, K, Cin, Cout, M, P, _pm), ts in agg.items():
    t = sorted(ts)[len(ts) // 2]
    rows.append((i, K, Cin, Cout, M, P / (K * M), t, 2.0 * P * Cin * Cout / t / 1e6, (4.0 * P * Cin + 4 * M * Cout + 8 * P + 4 * K * Cin * Cout) / t / 1e3))
tot = sum(r[6] for r in rows)
print(f"launches {len(rows)} total {tot/1e3:.2f} ms")
groups = collections.OrderedDict()
for r in rows:
    groups.setdefault(r[1:5], []).append(r)
print("K Cin Cout M | n dens us_each us_total share TF/s GB/s")
for key, rs in sorted(groups.items(), key=lambda kv: -sum(r[6] for r in kv[1])):
    n = len(rs); tt = sum(r[6] for r in rs)
    if tt / tot < 0.004: continue
    print(*key, "|", n, f"{rs[0][5]:.2f} {tt/n:.0f} {tt:.0f} {100*tt/tot:.1f}% {sum(r[7]*r[6] for r in rs)/tt:.1f} {sum(r[8]*r[6] for r in rs)/tt:.0f}")
