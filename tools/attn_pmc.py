"""Attention kernels alone, for `rocprofv3 --pmc` (the program goes directly after `--`): the decoder's superpoint cross-attention
(Q x 3000 keys, 8 heads, [content | positional] 64-channel scores, bit mask) at Q = 200 (headline) and Q = 3000 (query_num = -1),
fp32 and bf16 contractions, and the self-attention at Q = 200.  Prints the HIP-event time per call for the same launches."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from segdino3d_amd import ops

d = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
S, H, REPS = 3000, 8, 20
QS = [int(a) for a in sys.argv[1:]] or [200, 3000]          # one query count per profiled run keeps the per-kernel counter sums per shape
for Q in QS:
    qc, qs = torch.randn(Q, 256, generator=g).to(d), torch.randn(Q, 256, generator=g).to(d)
    kc, kp, v = (torch.randn(S, 256, generator=g).to(d) for _ in range(3))
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (Q, (S + 31) // 32), generator=g, dtype=torch.int64).to(torch.int32).to(d)
    for mode in ("fp32", "bf16"):
        with ops.bf16_decoder_scope(mode == "bf16"):
            for _ in range(3):
                ops.attention(qc, kc, v, H, 64 ** -0.5, mask_bits=bits, q2=qs, k2=kp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                ops.attention(qc, kc, v, H, 64 ** -0.5, mask_bits=bits, q2=qs, k2=kp)
            e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / REPS
        fl = 2.0 * Q * S * (64 + 32) * H
        print(f"cross-attention Q={Q} S={S} {mode}: {us:.1f} us per call = {fl / us / 1e6:.1f} TFLOP/s")
if 200 not in QS:
    sys.exit(0)
q = torch.randn(200, 768, generator=g).to(d)
for _ in range(3):
    ops.attention(q[:, :256], q[:, 256:512], q[:, 512:], H, 32 ** -0.5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    ops.attention(q[:, :256], q[:, 256:512], q[:, 512:], H, 32 ** -0.5)
e1.record(); torch.cuda.synchronize()
print(f"self-attention Q=200: {1e3 * e0.elapsed_time(e1) / REPS:.1f} us per call")
