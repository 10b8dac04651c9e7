"""Shared by the oracle (CPU) and device (GPU) gradient tests of the decoder: the objective of
tests/golden/make_golden_decoder_grad.py and the comparison against its fixture (reference autograd)."""
import os

import numpy as np
import torch

from _det import det_randn

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decoder_grad_s96_q16.npz"))


def objective(sets):
    """sets: list of dicts (aux layers first, final last) with tensors cls_preds / masks / centers / sizes / sem_preds or None."""
    total = 0.0
    for li, o in enumerate(sets):
        for k in ("cls_preds", "masks", "centers", "sizes", "sem_preds"):
            v = o.get(k)
            if v is None:
                continue
            total = total + (v * det_randn(f"gradw.{k}{li}", tuple(v.shape)).to(v.device, v.dtype)).sum()
    return total


def compare(named_grads, dx, dq, obj, tol):
    """named_grads: {parameter name: gradient}.  Every parameter the reference differentiates must agree in norm and in its
    first 24 entries (relative to the norm); parameters the reference leaves without gradient must have none / zero."""
    assert abs(float(obj) - float(Z["objective"])) <= 1e-4 * abs(float(Z["objective"]))
    ref_names = {k[len("norm/"):] for k in Z.files if k.startswith("norm/")}
    # key biases shift every score of a query by the same amount, which softmax ignores: their true gradient is zero and the
    # reference holds rounding noise there (norm ~1e-9) - errors are measured against at least 1e-2 of the median norm
    floor = 1e-2 * float(np.median([float(Z["norm/" + n]) for n in ref_names]))
    worst = []
    for name in sorted(ref_names):
        g = named_grads.get(name)
        assert g is not None, f"no gradient for {name}"
        g = g.detach().cpu().double().reshape(-1)
        n_ref = float(Z["norm/" + name])
        head = torch.from_numpy(Z["head/" + name]).double()
        scale = max(n_ref, floor)
        e_norm = abs(float(g.norm()) - n_ref) / scale
        e_head = float((g[: head.numel()] - head).abs().max()) / scale
        worst.append((max(e_norm, e_head), name))
    worst.sort(reverse=True)
    assert worst[0][0] <= tol, f"parameter gradients differ from the reference's: {worst[:6]}"
    for name, g in named_grads.items():
        if name not in ref_names and g is not None:
            assert float(g.abs().max()) == 0.0, f"{name} has a gradient the reference does not produce"
    for got, key in ((dx, "dx"), (dq, "dq")):
        ref = torch.from_numpy(Z[key]).double()
        err = float((got.detach().cpu().double() - ref).abs().max())
        assert err <= tol * float(ref.abs().max()), (key, err)
    return worst[0]
