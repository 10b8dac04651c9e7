#!/usr/bin/env python3
"""Golden vector for the decoder options `num_semantic_queries > 0` (learned query embeddings, prepended to the projected
superpoint queries, `instance_seg_3d_decoder.py:229-231, 300-308`) and `objectness_flag=True` (`out_score` head, `:263-265,
:548-550`) on the non-positional decoder variant: the REFERENCE decoder's outputs.

    python tests/golden/make_golden_plain_obj.py          (build container only: imports /root/reference)"""
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
    kw = dict(G.DECODER_KW_PLAIN, num_semantic_queries=7, objectness_flag=True)
    torch.manual_seed(0)
    dec = dec_mod.ScanNetQueryDecoder(**kw).eval()
    G.assign_det_weights(dec, "decoder.")
    x = G.decoder_inputs("decoder_plain_s40", 40, 3, 96)[0]          # same inputs as the plain fixture
    dec.return_hidden_states = True
    dec.return_aux_outputs = True
    with torch.no_grad():
        out = dec([x], None, None, [x], None, None, None, None)
    arrays = dict(x=x, cls_preds=out["cls_preds"][0], sem_preds=out["sem_preds"][0], masks=out["masks"][0], scores=out["scores"][0],
                  hidden_states=out["hidden_states"][0])
    for li, aux in enumerate(out["aux_outputs"]):
        if li in (0, 2, 4):
            arrays[f"aux{li}_cls"] = aux["cls_preds"][0]
            arrays[f"aux{li}_masks"] = aux["masks"][0]
            arrays[f"aux{li}_scores"] = aux["scores"][0]
    assert out["masks"][0].shape == (47, 40) and out["scores"][0].shape == (47, 1)
    G.save("decoder_plain_obj_s40", **arrays)


if __name__ == "__main__":
    main()
