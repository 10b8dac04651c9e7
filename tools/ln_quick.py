import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
d = torch.device("cuda:0")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for M, Cin in ((200, 256), (200, 1024)):
    x = torch.randn(M, Cin, device=d); w = torch.randn(256, Cin, device=d) * Cin ** -0.5; b = torch.randn(256, device=d)
    r = torch.randn(M, 256, device=d); g = torch.ones(256, device=d); be = torch.zeros(256, device=d)
    ops.LINEAR_LN_MAX_ROWS = 512
    tf = timeit(lambda: ops.linear_layernorm(x, w, b, g, be, res=r))
    ops.LINEAR_LN_MAX_ROWS = 0
    t2 = timeit(lambda: ops.linear_layernorm(x, w, b, g, be, res=r))
    print(f"M={M} Cin={Cin}: fused {tf:.1f} us, two launches {t2:.1f} us")
