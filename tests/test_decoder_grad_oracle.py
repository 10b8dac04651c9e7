"""oracle/decoder_ref.py differentiated by torch autograd against the gradients of the reference decoder itself
(tests/golden/decoder_grad_s96_q16.npz, reference autograd in train mode)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from decoder_grad_case import Z, compare, objective  # noqa: E402
from test_oracle_golden import decoder_state_dict, load  # noqa: E402


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 2e-4), (torch.float32, 2e-3)])
def test_oracle_decoder_gradients_match_reference(dtype, tol):
    from oracle import decoder_ref as D
    g = load("decoder_s96_q16")
    sd = {k: (v.to(dtype).requires_grad_(True) if v.is_floating_point() else v) for k, v in decoder_state_dict().items()}
    ids = g["query_ids"].long()
    x = g["x"].detach().clone().to(dtype).requires_grad_(True)
    q = g["x"].detach()[ids].clone().to(dtype).requires_grad_(True)
    t = lambda a: a.to(dtype)
    out = D.decoder_forward(sd, D.DecoderCfg(), x, t(g["pos"]), t(g["pos_wo"]), q, t(g["pos"][ids]), t(g["q2d_feat"]), t(g["q2d_pos"]),
                            t(g["lo"]), t(g["hi"]))
    assert torch.equal(out["masks"].detach() > 0, torch.from_numpy(Z["masks"]) > 0), "mask signs differ: gradients are not comparable"
    sets = [dict(a) for a in out["aux"]] + [dict(cls_preds=out["cls_preds"], masks=out["masks"], centers=out["centers"], sizes=out["sizes"],
                                                 sem_preds=out["sem_preds"])]
    obj = objective(sets)
    obj.backward()
    grads = {k[len("decoder."):]: v.grad for k, v in sd.items() if v.is_floating_point()}
    compare(grads, x.grad, q.grad, obj.detach(), tol)
