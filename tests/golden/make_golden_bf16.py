#!/usr/bin/env python3
"""Golden vector for BASELINE configs[2] (bf16 decoder): the REFERENCE decoder under `torch.autocast("cpu", bfloat16)` - what
`engine/train_engine_3d.py:88-100` does with `cfg.amp` - next to its own fp32 output on the same inputs and weights.

    python tests/golden/make_golden_bf16.py          (build container only: imports /root/reference)

The fixture pins the bf16 mode of the HIP decoder to the reference: tests/test_gpu_bf16_decoder.py asks that the HIP bf16
outputs are as close to the reference's autocast outputs as those are to the reference's own fp32 outputs (the intrinsic
bf16 noise of this network on this input), layer by layer.  Inputs and outputs only; weights come from key names (_det)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def main():
    G.install_stand_ins()
    sys.path.insert(0, G.REFERENCE)
    import segdino3d as seg  # noqa: F401 - the reference package
    from segdino3d.models.decoder import instance_seg_3d_decoder as dec_mod
    name, kw, S, M, nq = "decoder_bf16_s500_q32", G.DECODER_KW_SCANNET200, 500, 40, 32
    torch.manual_seed(0)
    dec = dec_mod.ScanNetQueryDecoder(**kw).eval()
    G.assign_det_weights(dec, "decoder.")
    x, pos, pos_wo, q2d_feat, q2d_pos, lo, hi = G.decoder_inputs("decoder_s500_q32", S, M, kw["in_channels"])   # same inputs as the fp32 fixture
    ids = torch.arange(0, S, S // nq)[:nq]
    q, qpos = x[ids], pos[ids]
    dec.return_hidden_states = True
    dec.return_aux_outputs = True
    arrays = dict(query_ids=ids)
    for tag, ctx in (("fp32", torch.autocast("cpu", enabled=False)), ("amp", torch.autocast("cpu", dtype=torch.bfloat16))):
        with torch.no_grad(), ctx:
            out = dec([x], [pos], [pos_wo], [q], [qpos], [q2d_feat], [q2d_pos.clone()], [(lo, hi)])
        for k in ("cls_preds", "sem_preds", "masks", "centers", "sizes", "hidden_states"):
            arrays[f"{tag}_{k}"] = out[k][0].float()
        for li, aux in enumerate(out["aux_outputs"]):
            if li in (0, 3):                                   # before any attention layer, and mid-way (keeps the fixture small)
                arrays[f"{tag}_aux{li}_masks"] = aux["masks"][0].float()
                arrays[f"{tag}_aux{li}_cls"] = aux["cls_preds"][0].float()
        print(tag, "dtypes:", out["masks"][0].dtype, out["cls_preds"][0].dtype)
    G.save(name, **arrays)
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    for k in ("cls_preds", "masks", "centers", "sizes"):
        print(f"  reference autocast vs reference fp32, {k}: rel L2 {rel(arrays['amp_' + k], arrays['fp32_' + k]):.4f}")
    bits = ((arrays["amp_masks"] > 0) == (arrays["fp32_masks"] > 0)).float().mean().item()
    print(f"  mask bits equal: {bits:.4f}")


if __name__ == "__main__":
    main()
