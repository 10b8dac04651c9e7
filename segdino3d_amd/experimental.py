"""Binding of libsegdino3d_hip_experimental.so (include/segdino3d_hip_experimental.h): kernels that are NOT part of the product
library - built and parity-tested, measured slower than the product path (profiles/EXPERIMENTS.md).  Nothing under
segdino3d_amd/ imports this module; tests marked `experimental` and the tools/slab_* scripts do.
Build: `make -C segdino3d_amd/csrc experimental` (`__graft_entry__.build()` does it too)."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, ops

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsegdino3d_hip_experimental.so")
_p, _i, _l, _z = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
SIGNATURES = {
    "sd3d_slab_conv_ws_bytes": (_z, [_i, _i, _i, _l, _l]),
    "sd3d_slab_conv": (_i, [_p, _i, _i, _p, _i, _p, _l, _p, _i, _i, _i, _l, _p, _p, _p, _i, _p, _i, _i, _p, _z, _p]),
}
_exp = None
_WS = ops._PerThread()       # slab_conv: per-workgroup rulebook scratch (+ partial slabs of the offset split)


def available() -> bool:
    return os.path.exists(LIB_PATH)


def load():
    global _exp
    if _exp is None:
        _lib.load()                                             # the product library first: the experimental one links against it
        if not available():
            raise _lib.HipExtensionMissing(f"{LIB_PATH} not found - build it with `make -C segdino3d_amd/csrc experimental`")
        lib = C.PyDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _exp = lib
    return _exp


def slab_conv_supported(K, Cin, Cout, M, n_pairs) -> bool:
    return load().sd3d_slab_conv_ws_bytes(K, Cin, Cout, M, int(n_pairs)) > 0


def slab_conv(x, wt, nbr, n_pairs=None, x2=None, scale=None, shift=None, res=None, act=None, out=None):
    """Same contract as gather_gemm(x, wt, nbr=...): output-stationary sparse convolution straight from the neighbour
    table (csrc/experimental/slab_conv.hip).  `n_pairs` (entries >= 0 of `nbr`) only guides the launch geometry."""
    lib = load()
    K, Cout, Cin = wt.shape
    if nbr.shape[0] != K:
        raise ValueError(f"weights have {K} offsets, the neighbour table {nbr.shape[0]}")
    M = nbr.shape[1]
    p0, ld0 = ops._rows(x, "x")
    C0 = x.shape[1]
    p1, ld1 = (None, 0)
    if x2 is not None:
        p1, ld1 = ops._rows(x2, "x2")
        if C0 + x2.shape[1] != Cin:
            raise ValueError(f"concat channels {C0}+{x2.shape[1]} != Cin {Cin}")
    elif C0 != Cin:
        raise ValueError(f"input channels {C0} != Cin {Cin}")
    if out is None:
        out = torch.empty(M, Cout, dtype=torch.float32, device=x.device)
    po, ldo = ops._rows(out, "out")
    pr, ldr = (None, 0)
    if res is not None:
        pr, ldr = ops._rows(res, "res")
    if n_pairs is None:
        n_pairs = K * M // 2
    nb = lib.sd3d_slab_conv_ws_bytes(K, Cin, Cout, M, int(n_pairs))
    if nb == 0:
        raise ValueError(f"slab_conv: shape K={K} Cin={Cin} Cout={Cout} is not supported")
    ws = _WS.get(nb, x.device)
    hook = ops.GG_HOOK
    if hook is not None:
        hook.before(dict(K=K, Cin=Cin, Cout=Cout, M=M, nbr=nbr, slab=True))
    _lib.check(lib.sd3d_slab_conv(p0, ld0, C0, p1, ld1, ops._ptr(nbr, torch.int32, "nbr"), int(n_pairs), ops._ptr(wt, torch.float32, "wt"),
                                  K, Cin, Cout, M, ops._ptr(scale, torch.float32, "scale"), ops._ptr(shift, torch.float32, "shift"), pr, ldr,
                                  po, ldo, ops.ACT[act], ws.data_ptr(), ws.numel(), ops._stream()), "slab_conv")
    if hook is not None:
        hook.after()
    return out
