"""Decoder projection products at training shapes (2441 rows = the query subset of a 3000-superpoint scene): fp32 kernels vs the
bf16-operand kernel, for the forward / input-gradient product (rows x Cin x Cout) and the weight-gradient product."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops, train_ops
from segdino3d_amd.train_dec import _identity_pairs


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


d = torch.device("cuda:0")
for M, cin, cout in ((2441, 256, 256), (2441, 256, 1024), (2441, 1024, 256), (2441, 512, 256), (3000, 256, 768), (200, 256, 256)):
    x = torch.randn(M, cin, device=d); w = torch.randn(cout, cin, device=d) * cin ** -0.5; g = torch.randn(M, cout, device=d)
    ws = ops.split_weights(w.unsqueeze(0), 1)
    t32 = timeit(lambda: ops.gather_gemm(x, w, exact=True))
    t16 = timeit(lambda: ops.gather_gemm(x, w, exact=True, wt_split=ws))
    tsp = timeit(lambda: ops.split_weights(w.unsqueeze(0), 1))
    pairs = _identity_pairs(M, d)
    tw32 = timeit(lambda: train_ops.pair_wgrad(g, x, pairs))
    tw16 = timeit(lambda: train_ops.pair_wgrad(g, x, pairs, bf16_operands=True))
    fl = 2.0 * M * cin * cout
    print(f"M={M} {cin}->{cout}: forward fp32 {t32:6.1f} us ({fl / t32 / 1e6:5.1f} TF) | bf16 {t16:6.1f} us ({fl / t16 / 1e6:5.1f} TF) + weight rounding {tsp:5.1f} us |"
          f" dW fp32 {tw32:6.1f} us | dW bf16-rounded operands {tw16:6.1f} us")
