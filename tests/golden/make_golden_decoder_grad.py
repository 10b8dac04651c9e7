"""Generates tests/golden/decoder_grad_s96_q16.npz: gradients of the REFERENCE decoder
(/root/reference/segdino3d/models/decoder/instance_seg_3d_decoder.py, ScanNet200 prototype, train mode) from torch
autograd, for the inputs and deterministic weights of the existing forward fixture decoder_s96_q16.
Scalar objective: sum over the final and all auxiliary prediction sets of <output, R> with R = det_randn("gradw.<key><layer>").
Stored: the objective, d/d(superpoint features), d/d(query features) in full, and for every parameter the gradient's L2
norm plus its first 24 entries (the full set is ~8 M floats).  Runs in the build container only.
`--amp` writes decoder_grad_amp_s96_q16.npz instead: the same run under torch.autocast("cpu", bfloat16) around the decoder
call (train_engine_3d.py:88-100, BASELINE configs[4]) - the yardstick for the bf16 training path's gradients."""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from _det import det_randn  # noqa: E402


def objective(out):
    total = 0.0
    sets = list(out["aux_outputs"]) + [out]
    for li, o in enumerate(sets):
        for k in ("cls_preds", "masks", "centers", "sizes", "sem_preds"):
            v = o.get(k)
            if v is None or v[0] is None:
                continue
            total = total + (v[0] * det_randn(f"gradw.{k}{li}", tuple(v[0].shape))).sum()
    return total


def main():
    G.install_stand_ins()
    sys.path.insert(0, G.REFERENCE)
    from segdino3d.models.decoder import instance_seg_3d_decoder as dec_mod
    name, S, M, Qn = "decoder_s96_q16", 96, 7, 16
    kw = G.DECODER_KW_SCANNET200
    torch.manual_seed(0)
    dec = dec_mod.ScanNetQueryDecoder(**kw).train()
    G.assign_det_weights(dec, "decoder.")
    x, pos, pos_wo, q2d_feat, q2d_pos, lo, hi = G.decoder_inputs(name, S, M, kw["in_channels"])
    ids = torch.arange(0, S, S // Qn)[:Qn]
    x = x.clone().requires_grad_(True)
    q = x.detach()[ids].clone().requires_grad_(True)
    dec.return_hidden_states = False
    dec.return_aux_outputs = True
    amp = "--amp" in sys.argv
    with (torch.autocast("cpu", dtype=torch.bfloat16) if amp else contextlib.nullcontext()):
        out = dec([x], [pos], [pos_wo], [q], [pos[ids]], [q2d_feat], [q2d_pos.clone()], [(lo, hi)])

    def as_f32(o):
        return {k: ([as_f32(a) for a in v] if k == "aux_outputs" else
                    v if not isinstance(v, (list, tuple)) else [t.float() if torch.is_tensor(t) else t for t in v]) for k, v in o.items()}

    out = as_f32(out)
    total = objective(out)
    total.backward()
    blob = dict(objective=total.detach().float().numpy(), dx=x.grad.float().numpy(), dq=q.grad.float().numpy(),
                masks=out["masks"][0].detach().float().numpy())
    n_used = 0
    for pname, p in dec.named_parameters():
        if p.grad is None:
            continue
        n_used += 1
        g = p.grad.reshape(-1).float()
        blob["norm/" + pname] = np.array(float(g.norm()))
        blob["head/" + pname] = g[:24].numpy().copy()
    path = os.path.join(HERE, "decoder_grad_amp_s96_q16.npz" if amp else "decoder_grad_s96_q16.npz")
    np.savez_compressed(path, **blob)
    print("objective", float(total), "parameters with gradient", n_used, "->", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
