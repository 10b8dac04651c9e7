"""Weight gradient of a Linear at the decoder's training shapes: the one-launch kernel (sd3d_linear_wgrad) against the pair-list kernel +
reduce on identity lists (train_ops.pair_wgrad), us per call."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import _lib, ops, train_ops
lib = _lib.load()
d = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


g_ = torch.Generator().manual_seed(0)
print("| M | Cin | Cout | one launch us (TFLOP/s) | pair lists + reduce us (TFLOP/s) |")
print("|---:|---:|---:|---:|---:|")
for M, cin, cout in [(2441, 256, 256), (2441, 256, 1024), (2441, 1024, 256), (2441, 256, 768), (2441, 512, 256), (3000, 256, 3072), (3000, 256, 1536), (3000, 96, 256),
                     (301, 256, 3072), (2441, 256, 201), (2441, 256, 3)]:
    c_pad = (cout + 31) // 32 * 32
    g = torch.zeros(M, c_pad); g[:, :cout] = torch.randn(M, cout, generator=g_)
    x = torch.randn(M, cin, generator=g_)
    g, x = g.to(d), x.to(d)
    dw, db = torch.empty(cout, cin, device=d), torch.empty(cout, device=d)
    t1 = timeit(lambda: lib.sd3d_linear_wgrad(g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), M, cin, cout, dw.data_ptr(), db.data_ptr(), 0, ops._stream()))
    nbr = torch.arange(M, dtype=torch.int32, device=d).unsqueeze(0).contiguous()
    pl = ops.pair_lists(nbr, M)
    t2 = timeit(lambda: train_ops.pair_wgrad(g, x, pl, bias=True))
    fl = 2.0 * M * cin * cout
    print(f"| {M} | {cin} | {cout} | {t1:.1f} ({fl / t1 / 1e6:.1f}) | {t2:.1f} ({fl / t2 / 1e6:.1f}) |")
