"""Decoder in fp32 vs bf16 mode (BASELINE configs[1] / configs[2]) at the two query counts of the reference:
200 queries (headline) and one query per superpoint (query_num = -1, the parity configuration): whole-decoder time and
the superpoint cross-attention alone (Q x 3000 keys, 8 heads, [content | positional] 64-channel scores)."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import segdino3d_amd as seg
from segdino3d_amd import ops
from segdino3d_amd.configs import scannet200_model_cfg

d = torch.device("cuda:0")
torch.manual_seed(0)
model = seg.build_architecture(scannet200_model_cfg(query_num=200)).eval().to(d)
dec = model.decoder
g = torch.Generator().manual_seed(0)
S, M = 3000, 300
pos = (torch.rand(S, 3, generator=g) * torch.tensor([8.0, 6.0, 3.0])).to(d)
x = torch.randn(S, 96, generator=g).to(d)
q2d_feat, q2d_pos = torch.randn(M, 256, generator=g).to(d), pos[:M].clone()
lo, hi = pos.min(0)[0] - 0.03, pos.max(0)[0] + 0.05


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
for Q in (200, 3000):
    ids = torch.arange(0, S, S // Q)[:Q].to(d)
    args = ([x], [pos], [pos], [x[ids]], [pos[ids]], [q2d_feat], [q2d_pos], [(lo, hi)])
    qc, qs = torch.randn(Q, 256, generator=g).to(d), torch.randn(Q, 256, generator=g).to(d)
    kc, kp, v = (torch.randn(S, 256, generator=g).to(d) for _ in range(3))
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (Q, (S + 31) // 32), generator=g, dtype=torch.int64).to(torch.int32).to(d)
    row = {}
    for mode in ("fp32", "bf16"):
        dec.compute_dtype = mode
        with torch.no_grad():
            row[f"decoder_ms_{mode}"] = round(timeit(lambda: dec(*args)), 3)
        with ops.bf16_decoder_scope(mode == "bf16"):
            us = 1e3 * timeit(lambda: ops.attention(qc, kc, v, 8, 64 ** -0.5, mask_bits=bits, q2=qs, k2=kp), 20)
        flops = 2.0 * Q * S * (64 + 32) * 8
        row[f"cross_attention_us_{mode}"] = round(us, 1)
        row[f"cross_attention_TFLOPs_{mode}"] = round(flops / us / 1e6, 2)
    dec.compute_dtype = "fp32"
    out[f"Q={Q}"] = row
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/decoder_dtype.json", "w"), indent=1)
