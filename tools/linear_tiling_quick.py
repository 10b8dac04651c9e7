"""Dense Linear products (ops.gather_gemm with identity rows) at the decoder's mid-size row counts, every tiling the launcher
knows: auto (0), split contraction (-1), lock-step LDS with 1-4 column tiles (-11..-14), private fragments with 1-4 column tiles.
    python tools/linear_tiling_quick.py [rows ...]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops

d = torch.device("cuda:0")
rows = [int(a) for a in sys.argv[1:]] or [1000, 2048, 2441, 3000, 4096]
shapes = [(256, 256), (256, 768), (256, 1024), (1024, 256), (512, 256), (96, 256), (256, 3072), (256, 199)]
codes = [0, -1, -11, -12, -14, 1, 2, 4]


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


print("us per launch (back-to-back launches on one stream); code 0 = what the launcher picks")
print(f"{'rows':>6} {'Cin':>5} {'Cout':>5} | " + " ".join(f"{c:>7}" for c in codes) + " | picked")
for M in rows:
    for cin, cout in shapes:
        x = torch.randn(M, cin, device=d)
        w = torch.randn(1, cout, cin, device=d) * cin ** -0.5
        b = torch.randn(cout, device=d)
        ts = []
        for c in codes:
            if c > 0 and ((cout + 31) // 32) % c:
                ts.append(float("nan"))
                continue
            try:
                ts.append(timed(lambda: ops.gather_gemm(x, w, shift=b, nt=c, exact=True)))
            except Exception:
                ts.append(float("nan"))
        print(f"{M:>6} {cin:>5} {cout:>5} | " + " ".join(f"{t:7.1f}" for t in ts) + f" | {ops.dense_code(M, cin, cout)}")
