"""Per-op cost of the row-chain executor (csrc/rowchain.hip) at the decoder's shapes: us per launch for programs that repeat one op.
usage: python tools/rowchain_quick.py [rows] [tile rows: 16 | 4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from segdino3d_amd import ops
from segdino3d_amd.rowchain import Program

d = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200
TILE = int(sys.argv[2]) if len(sys.argv) > 2 else 16          # rows per workgroup: 16 or 4
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, 256, generator=g).to(d)
ws = [(torch.randn(256, 256, generator=g) / 16).to(d) for _ in range(16)]
b = torch.randn(256, generator=g).to(d)
w_up, w_dn = (torch.randn(1024, 256, generator=g) / 16).to(d), (torch.randn(256, 1024, generator=g) / 32).to(d)
b_up = torch.randn(1024, generator=g).to(d)
w3 = (torch.randn(3, 256, generator=g) / 16).to(d)
out = torch.empty(rows, 256, device=d)
qkv = torch.randn(rows, 768, generator=g).to(d)
kv2d = torch.randn(301, 512, generator=g).to(d)
S = 3000
nw = (S + 31) // 32
blocked = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, nw), generator=g, dtype=torch.int64).to(torch.int32).to(d)
near = torch.randint(-2 ** 31, 2 ** 31 - 1, (300, nw), generator=g, dtype=torch.int64).to(torch.int32).to(d)
near = near & torch.randint(-2 ** 31, 2 ** 31 - 1, (300, nw), generator=g, dtype=torch.int64).to(torch.int32).to(d)
scene = [dict(q0=0, nq=rows, m0=0, nm=301, bits_off=0, nw=nw, near_off=0)]


def timeit(build, n=30, **kw):
    P = build()                                                # built once: the loop below measures the GPU, not the host

    def once():
        P.launch(scene, **kw)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        once()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def prog_linears(k):
    def f():
        P = Program(8, rows=TILE)
        P.load(0, x)
        for i in range(k):
            P.linear(1 if i % 2 == 0 else 0, 0 if i % 2 == 0 else 1, ws[i % 16], b, act="relu")
        P.store(0, out)
        return P
    return f


print(f"rows = {rows} ({(rows + TILE - 1) // TILE} workgroups of {16 if TILE == 16 else 8} waves, {TILE} rows each)")
t0 = timeit(prog_linears(0))
print(f"LOAD + STORE only: {t0:.1f} us per launch")
for k in (1, 2, 4, 8, 16, 32):
    t = timeit(prog_linears(k))
    print(f"{k:2d} x LINEAR 256 -> 256: {t:.1f} us per launch, {(t - t0) / k:.2f} us per op")


def prog_same_w(k):
    def f():
        P = Program(8, rows=TILE)
        P.load(0, x)
        for i in range(k):
            P.linear(1 if i % 2 == 0 else 0, 0 if i % 2 == 0 else 1, ws[0], b, act="relu")
        P.store(0, out)
        return P
    return f


t = timeit(prog_same_w(16))
print(f"16 x LINEAR 256 -> 256 with the SAME weights (L2 / L1 warm): {t:.1f} us, {(t - t0) / 16:.2f} us per op")


def prog_ffn(k):
    def f():
        P = Program(8, rows=TILE)
        P.load(0, x)
        for i in range(k):
            P.linear(4, 0, w_up, b_up, act="gelu")
            P.linear(0, 4, w_dn, b)
        P.store(0, out)
        return P
    return f


t = timeit(prog_ffn(4))
print(f"4 x (LINEAR 256 -> 1024 gelu, LINEAR 1024 -> 256): {t:.1f} us, {(t - t0) / 4:.2f} us per pair")


def prog_op(kind, k):
    def f():
        P = Program(8, rows=TILE)
        P.load(0, x)
        for i in range(k):
            if kind == "ln":
                P.ln(1, 0, b, b)
            elif kind == "lin3":
                P.linear(1, 0, w3, None)
            elif kind == "sa":
                P.attn(1, 0, qkv[:, 256:512], qkv[:, 512:], 0.17, aux=6)
            elif kind == "ca2d":
                P.attn(1, 0, kv2d[:, :256], kv2d[:, 256:], 0.17, aux=6, keys_2d=True, masked=True)
            elif kind == "bits2d":
                P.bits2d(blocked, near)
            elif kind == "load":
                P.load(1, x)
            elif kind == "store":
                P.store(0, out)
        P.store(0, out)
        return P
    return f


for kind in ("ln", "lin3", "sa", "ca2d", "bits2d", "load", "store"):
    t = timeit(prog_op(kind, 8), nw_max=nw, nw2_max=10)
    print(f"8 x {kind}: {t:.1f} us, {(t - t0) / 8:.2f} us per op")
# host cost of building + launching a 20-op program
t1 = time.perf_counter()
for _ in range(200):
    P = prog_linears(16)()
    P.launch(scene)
torch.cuda.synchronize()
print(f"host: {1e6 * (time.perf_counter() - t1) / 200:.0f} us per build + launch of an 18-op program (GPU time included if larger)")
