"""top-k of the post-processing: radix select (one launch) against the full radix sort, us per call (HIP events, 200 calls)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for n, k in ((39_600, 600), (40_960, 1024), (8_000, 600)):
    x = (torch.sigmoid(torch.randn(n, generator=g) * 3) * torch.sigmoid(torch.randn(n, generator=g))).to(d)
    def sel(): return ops.topk_desc(x, k)
    def srt(): return ops.sort_pairs(ops.keys_from_f32(x, descending=True), None, 0, 32)[1][:k]
    for name, f in (("select", sel), ("sort", srt)):
        for _ in range(20): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(200): f()
        e1.record(); torch.cuda.synchronize()
        print(f"n = {n}, k = {k}: {name:6s} {1e3 * e0.elapsed_time(e1) / 200:7.1f} us per call")
