"""Dense Linear launch shapes of the decoder: time per tiling code of sd3d_gather_gemm (0 = heuristic)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

for M, cin, cout in [(3000, 256, 256), (3000, 256, 1024), (3000, 1024, 256), (3000, 96, 256), (3000, 256, 3072), (200, 256, 256), (200, 256, 1024), (200, 1024, 256)]:
    x = torch.randn(M, cin, generator=g).to(d); w = torch.randn(cout, cin, generator=g).to(d).unsqueeze(0).contiguous()
    row = []
    for nt in (0, 1, 2, 4, -1, -11, -12, -14):
        try:
            t = timeit(lambda: ops.gather_gemm(x, w, nt=nt))
            row.append(f"{nt}:{t:.1f}")
        except Exception as e:  # noqa: BLE001
            row.append(f"{nt}:err")
    print(f"M={M} {cin}->{cout}  " + "  ".join(row))
    if M <= 512:                                               # the decoder's shape: bias, and bias + residual
        b, r = torch.randn(cout, generator=g).to(d), torch.randn(M, cout, generator=g).to(d)
        print(f"      with bias: {timeit(lambda: ops.gather_gemm(x, w, shift=b)):.1f} us, bias + residual + relu: "
              f"{timeit(lambda: ops.gather_gemm(x, w, shift=b, res=r, act='relu')):.1f} us")
