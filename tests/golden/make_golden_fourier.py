#!/usr/bin/env python3
"""Golden vector for the decoder option `pos_type="fourier"` (Gaussian random Fourier features,
`segdino3d/models/module/utils.py:107-142`): the REFERENCE decoder with the ScanNet200 kwargs, `pos_type="fourier"`,
`box_modulate_ca=False` (the reference asserts sine for box modulation, `instance_seg_3d_decoder.py:528`).

    python tests/golden/make_golden_fourier.py          (build container only: imports /root/reference)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def main():
    G.install_stand_ins()
    sys.path.insert(0, G.REFERENCE)
    import segdino3d as seg  # noqa: F401 - the reference package
    from segdino3d.models.decoder import instance_seg_3d_decoder as dec_mod
    kw = dict(G.DECODER_KW_SCANNET200, pos_type="fourier", box_modulate_ca=False, gauss_scale=1.0)
    dec = G.golden_decoder(dec_mod, "decoder_fourier_s48", kw, S=48, M=5)
    assert "position_embedding.gauss_B" in dec.state_dict() and "ref_anchor_head.layers.0.weight" not in dec.state_dict()


if __name__ == "__main__":
    main()
