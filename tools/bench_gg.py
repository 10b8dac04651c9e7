#!/usr/bin/env python3
"""Micro-benchmark of the gather-GEMM variants on the real kernel maps of one synthetic scene.

    python tools/bench_gg.py [--points 150000] [--reps 5]

For each representative layer shape of Res16UNet34C it times every kernel variant (nt codes of
sd3d_gather_gemm: 1..4 private fragments, -11..-14 LDS-shared weights, 0 heuristic) with HIP events and
prints us / dense-equivalent TFLOP/s / active TFLOP/s.  Results are checked against variant nt=1.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdino3d_amd import ops  # noqa: E402
from segdino3d_amd.sparse import SceneMaps  # noqa: E402
from segdino3d_amd.synth import make_scene  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=150000)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    d = torch.device("cuda:0")
    pts, tgt = make_scene(0, args.points, 3000, 300)
    maps = SceneMaps(pts.to(d), 0.02, 5, superpoints=tgt.extra_features["super_point_masks"].to(d))
    print("voxels per level", maps.n_vox)
    g = torch.Generator().manual_seed(0)
    shapes = [  # (name, level, kind, Cin, Cout)
        ("stem k5 288->32 L0", 0, ("same", 5), 288, 32),
        ("block8 k3 96->96 L0", 0, ("same", 3), 96, 96),
        ("block8 k3 128->96 L0", 0, ("same", 3), 128, 96),
        ("block7 k3 96->96 L1", 1, ("same", 3), 96, 96),
        ("block6 k3 128->128 L2", 2, ("same", 3), 128, 128),
        ("block2 k3 64->64 L2", 2, ("same", 3), 64, 64),
        ("block5 k3 256->256 L3", 3, ("same", 3), 256, 256),
        ("block3 k3 128->128 L3", 3, ("same", 3), 128, 128),
        ("block4 k3 256->256 L4", 4, ("same", 3), 256, 256),
        ("down k2 32->32 L0->1", 0, ("down",), 32, 32),
        ("up k2 96->96 L1->0", 0, ("up",), 96, 96),
    ]
    variants = [0, -11, -13, -21, -22, -23, -31, -32, -33]
    for name, lvl, kind, cin, cout in shapes:
        if kind[0] == "same":
            nbr = maps.same(lvl, kind[1]); vin = maps.n_vox[lvl]
        elif kind[0] == "down":
            nbr = maps.down(lvl); vin = maps.n_vox[lvl]
        else:
            nbr = maps.up(lvl); vin = maps.n_vox[lvl + 1]
        K, M = nbr.shape
        P = int((nbr >= 0).sum())
        x = torch.randn(vin, cin, generator=g).to(d)
        w = (torch.randn(K, cout, cin, generator=g) * (K * cin) ** -0.5).to(d)
        ref = None
        row = []
        for nt in variants:
            sub = (cout + 31) // 32
            eff = nt if nt > 0 else (-nt - 30 if nt <= -31 else (-nt - 20 if nt <= -21 else (-nt - 10 if nt <= -11 else None)))
            if eff is not None and (eff > sub or sub % eff):
                row.append("   -   ")
                continue
            fn = lambda: ops.gather_gemm(x, w, nbr=nbr, act="relu", nt=nt)  # noqa: E731
            try:
                us = timeit(fn, args.reps)
            except RuntimeError as e:
                row.append(" err ")
                continue
            out = fn()
            if ref is None:
                ref = out
            else:
                err = (out - ref).abs().max().item()
                if err > 1e-3:
                    row.append(f"BAD{err:.0e}")
                    continue
            row.append(f"{us:7.0f}")
        dense = 2.0 * M * K * cin * cout
        act = 2.0 * P * cin * cout
        print(f"{name:26s} M={M:6d} K={K:3d} P/KM={P / (K * M):.2f} | " + " ".join(f"{v:>7}" for v in variants))
        print(f"{'':26s} dense {dense / 1e9:6.1f} GF active {act / 1e9:6.1f} GF      | " + " ".join(row))
    # decoder-like linears
    for (M, cin, cout) in [(200, 256, 256), (200, 256, 1024), (200, 1024, 256), (3000, 256, 3072), (3000, 96, 256)]:
        x = torch.randn(M, cin, generator=g).to(d)
        w = (torch.randn(cout, cin, generator=g) * cin ** -0.5).to(d)
        row = []
        for nt in [0, 1, 2, 4, -1, -11, -12, -14]:
            fn = lambda: ops.gather_gemm(x, w, nt=nt)  # noqa: E731
            row.append(f"{nt}:{timeit(fn, 20):.1f}")
        print(f"linear {M}x{cin}->{cout}: " + "  ".join(row))


if __name__ == "__main__":
    main()
