"""Ceiling of the pass-1 microkernels: both pass-1 kernels of the sparse convolution with the IDENTITY as their
rulebook - no gather randomness, no partial products, every MFMA row real - at every layer shape of Res16UNet34C, on as many rows as the
layer has rulebook entries (same flops as the sparse layer).  If dense rows do not reach >= 0.70 of the fp32 matrix peak either, the
microkernel - not the lists, not the chains - is what bounds pass 1.

  lock-step        : `pair_dense_kernel_*` (launch_pair_dense: the lock-step kernel, identity rulebook built on the fly, direct epilogue)
  weight-stationary: `pair_gemm_ws_direct_kernel_*` on a K = 1 identity pair table (W staged once per workgroup range, no per-step barrier)
  ws, crowded      : the same table with `scenes_in_flight(4)` (-> the lock-step direct kernel reading the lists)
usage: python tools/dense_ceiling.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops  # noqa: E402

PEAK = 157.3


def timeit(fn, reps=7):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# (level, Cin, Cout, rulebook entries of the benchmark scene's table) - profiles/r05_pair_pool_ab.txt
shapes = [(0, 288, 32, 990230), (0, 96, 96, 423582), (0, 128, 96, 423582), (1, 32, 32, 777134), (1, 96, 96, 777134), (1, 128, 96, 777134),
          (2, 64, 64, 680872), (2, 128, 128, 680872), (2, 192, 128, 680872), (3, 128, 128, 227930), (3, 256, 256, 227930),
          (3, 384, 256, 227930), (4, 256, 256, 56881)]
print("| level | Cin -> Cout | rows | lock-step dense us (TF/s, frac) | weight-stationary identity us (TF/s, frac) | lock-step over identity lists us (TF/s, frac) |")
print("|---|---|---:|---:|---:|---:|")
for lvl, cin, cout, P in shapes:
    x = torch.randn(P, cin, generator=g).to(d)
    w = (torch.randn(1, cout, cin, generator=g) * cin ** -0.5).to(d)
    out = torch.empty(P, cout, device=d)
    flops = 2.0 * P * cin * cout
    t_ls = timeit(lambda: ops.gather_gemm(x, w, out=out))
    if os.environ.get("CEIL_ONLY_LS") == "1":
        print(f"| {lvl} | {cin} -> {cout} | {P} | {t_ls:.0f} ({flops / t_ls / 1e6:.1f}, {flops / t_ls / 1e6 / PEAK:.2f}) | | |")
        continue
    nbr = torch.arange(P, dtype=torch.int32, device=d).view(1, P)
    pl = ops.pair_lists(nbr, P, direct=True)
    t_ws = timeit(lambda: ops.pair_conv(x, w, pl, out=out))
    with ops.scenes_in_flight(4):
        t_lsl = timeit(lambda: ops.pair_conv(x, w, pl, out=out))
    f = lambda t: f"{t:.0f} ({flops / t / 1e6:.1f}, {flops / t / 1e6 / PEAK:.2f})"  # noqa: E731
    print(f"| {lvl} | {cin} -> {cout} | {P} | {f(t_ls)} | {f(t_ws)} | {f(t_lsl)} |")
    del x, w, out, nbr, pl
    torch.cuda.empty_cache()
