"""Attention backward at the training shapes (2441 queries x 3000 keys cross-attention with two score sources, 2441 x 2441 self-attention):
us per backward call (both kernels), HIP events.   python tools/attn_bwd_quick.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import train_dec as T
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for Lq, Lk, nsrc, masked in [(2441, 3000, 2, True), (2441, 2441, 1, False), (2441, 301, 1, True)]:
    q = torch.randn(Lq, 512, generator=g).to(d).requires_grad_(True)
    k = torch.randn(Lk, 768, generator=g).to(d).requires_grad_(True)
    bits = torch.randint(-2**31, 2**31 - 1, (Lq, (Lk + 31) // 32), dtype=torch.int32, generator=g).to(d) if masked else None
    dy = torch.randn(Lq, 256, generator=g).to(d)
    def run():
        out = T.attention(q[:, :256], k[:, :256], k[:, 512:], 8, 0.125, mask_bits=bits, q2=q[:, 256:] if nsrc == 2 else None, k2=k[:, 256:512] if nsrc == 2 else None)
        return out
    out = run()
    for _ in range(3):
        q.grad = k.grad = None
        out.backward(dy, retain_graph=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out.backward(dy, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    print(f"Lq {Lq} Lk {Lk} sources {nsrc} masked {masked}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us per backward (incl. autograd slicing / accumulation)")
