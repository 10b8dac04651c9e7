"""oracle/loss_ref.py against the reference's own criterion outputs (tests/golden/loss_criterion.npz)."""
import pytest
import torch

from oracle import loss_ref
from tests.loss_cases import KEYS, as_pred, load_case


@pytest.mark.parametrize("name", ["s200", "base", "hung"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_oracle_criterion_matches_reference(name, dtype):
    cfg, targets, layers, exp = load_case(name, dtype)
    for layer in layers:
        for k in KEYS:
            layer[k] = [None if v is None else v.requires_grad_(True) for v in layer[k]]
    out = loss_ref.unified_criterion(as_pred(layers), targets, cfg)
    assert abs(float(out["seg_loss"].detach()) - exp["seg_loss"]) < 2e-6 * max(1.0, abs(exp["seg_loss"]))
    assert abs(float(out["inst_loss"].detach()) - exp["inst_loss"]) < 3e-6 * abs(exp["inst_loss"])
    (out["seg_loss"] + out["inst_loss"]).backward()
    for l, layer in enumerate(layers):
        for k in KEYS:
            for b, v in enumerate(layer[k]):
                if v is None:
                    continue
                g = v.grad if v.grad is not None else torch.zeros_like(v)
                ref = exp["grads"][l][k][b]
                assert (g.float() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item()), (l, k, b)
