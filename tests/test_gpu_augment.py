"""Device augmentation (segdino3d_amd/augment.py + csrc/augment.hip) against the reference's own
transform outputs (tests/golden/augment.npz, same numpy.random seeds) and, at full size, through properties
(zero noise = identity, inverse affine round trip, blur conserves mass away from the border)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))


def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.mark.parametrize("ci", range(5))
def test_train_transform_matches_reference_outputs(ci):
    from segdino3d_amd.augment import Scannet200Transforms
    from segdino3d_amd.gtypes import GD3DTarget
    d = dev()
    pts = torch.from_numpy(Z[f"c{ci}/points_in"].copy()).to(d)
    tgt = GD3DTarget(extra_features={"query2d_pos": torch.from_numpy(Z[f"c{ci}/query2d_pos_in"].copy()).to(d)})
    np.random.seed(int(Z[f"c{ci}/seed"]))
    out, tgt = Scannet200Transforms("train", voxel_size=0.02)(pts, tgt)
    assert [tgt["pcd_horizontal_flip"], tgt["pcd_vertical_flip"]] == list(Z[f"c{ci}/flags"])
    assert abs(tgt["pcd_scale_factor"] - float(Z[f"c{ci}/scale"])) < 1e-12
    np.testing.assert_allclose(tgt["pcd_rotation"].numpy(), Z[f"c{ci}/rotation"], atol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), Z[f"c{ci}/points"], rtol=3e-6, atol=3e-6)
    np.testing.assert_allclose(tgt["extra_features"]["query2d_pos"].cpu().numpy(), Z[f"c{ci}/query2d_pos"], rtol=3e-6, atol=3e-6)
    # voxel units (hundreds): float32 resolution ~3e-5; the reference carries float64 between the two passes
    np.testing.assert_allclose(tgt["elastic_coords"].cpu().numpy(), Z[f"c{ci}/elastic_coords"], rtol=1e-6, atol=3e-4)
    np.testing.assert_allclose(tgt["extra_features"]["elastic_coords_query2d_pos"].cpu().numpy(), Z[f"c{ci}/elastic_coords_query2d_pos"],
                               rtol=1e-6, atol=3e-4)
    assert tgt["coords_voxel_size"] == 0.02


def test_val_transform_matches_reference_outputs():
    from segdino3d_amd.augment import Scannet200Transforms
    d = dev()
    pts = torch.from_numpy(Z["val/points_in"].copy()).to(d)
    out, _ = Scannet200Transforms("val")(pts, {"extra_features": {}})
    np.testing.assert_allclose(out.cpu().numpy(), Z["val/points"], rtol=2e-6, atol=2e-6)


def test_blurred_noise_matches_scipy():
    import scipy.ndimage
    from segdino3d_amd.augment import blurred_noise
    d = dev()
    np.random.seed(3)
    got = blurred_noise((9, 14, 5), device=d).cpu().numpy()
    np.random.seed(3)
    noise = [np.random.randn(9, 14, 5).astype("float32") for _ in range(3)]
    blurs = [np.ones(s, dtype="float32") / 3 for s in ((3, 1, 1), (1, 3, 1), (1, 1, 3))]
    for b in blurs + blurs:
        noise = [scipy.ndimage.convolve(n, b, mode="constant", cval=0) for n in noise]
    np.testing.assert_allclose(got, np.stack(noise), rtol=0, atol=1e-7)


def test_full_size_properties():
    """150 k points: the inverse affine restores the scene, zero noise moves nothing, points outside the noise grid are
    not moved, a constant noise volume moves every inside point by exactly mag * constant."""
    from segdino3d_amd import augment as A
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, _ = make_scene(3, 150_000, 3000, 50)
    pts = pts.to(d)
    ref = pts.clone()
    A.affine_(pts, flip_x=True, flip_y=False, angle=1.234, scale=1.1, trans=(0.3, -0.2, 0.05))
    assert (pts[:, :3] - ref[:, :3]).abs().max().item() > 0.5 and torch.equal(pts[:, 3:], ref[:, 3:])
    # inverse: undo translation and scale, rotate back, flip back (flip is applied first, so it is undone last)
    A.affine_(pts, trans=(-0.3, 0.2, -0.05))
    A.affine_(pts, scale=1 / 1.1)
    A.affine_(pts, angle=-1.234)
    A.affine_(pts, flip_x=True)
    assert (pts[:, :3] - ref[:, :3]).abs().max().item() < 5e-6
    el = A.ElasticTransfrom(gran=[6, 20], mag=[40, 160], voxel_size=0.02, p=1.0)
    coords = el._voxel_units(ref)
    assert np.array_equal(coords.cpu().numpy(), ref[:, :3].cpu().numpy() / 0.02)      # true fp32 division, as numpy does (`:417`)
    extent = coords.abs().amax(0).cpu().numpy()
    dims = extent.astype(np.int32) // 6 + 3
    zero = torch.zeros(3, *[int(v) for v in dims], device=d)
    moved = coords.clone()
    el._displace(moved, zero, 6, 40)
    assert torch.equal(moved, coords)
    const = torch.full_like(zero, 0.25)
    el._displace(moved, const, 6, 40)
    assert (moved - coords - 10.0).abs().max().item() < 1e-4                    # everything is inside the grid by construction
    small = torch.full((3, 2, 2, 2), 1.0, device=d)                             # grid [-6, 6]^3: most points lie outside
    moved = coords.clone()
    el._displace(moved, small, 6, 40)
    inside = (coords.abs() <= 6).all(dim=1)
    assert torch.equal(moved[~inside], coords[~inside]) and bool(((moved[inside] - coords[inside] - 40.0).abs() < 1e-4).all())


def test_cpu_tensors_are_refused():
    from segdino3d_amd import augment as A
    dev()
    with pytest.raises(RuntimeError):
        A.affine_(torch.zeros(4, 6), flip_x=True)


def _elastic_scene(n, S, idx):
    """A synthetic scene pushed through the device train transform (seeded so that the elastic distortion is applied)."""
    from segdino3d_amd.augment import Scannet200Transforms
    from segdino3d_amd.synth import make_scene
    d = dev()
    pts, tgt = make_scene(idx, n_points=n, n_superpoints=S, n_query2d=20)
    pts[:, 3:] = (pts[:, 3:] * 40 + 120).clamp(0, 255)               # raw colours, as the loader hands them over
    pts, tgt = pts.to(d), tgt.to(d)
    np.random.seed(5)                                                # seed 5: elastic applied (tests/golden/make_golden_aug.py)
    pts, tgt = Scannet200Transforms("train", voxel_size=0.02)(pts, tgt)
    assert (tgt["elastic_coords"] - pts[:, :3] / 0.02).abs().max().item() > 5.0
    return pts, tgt


@pytest.mark.parametrize("backbone", ["mink", "spconv"])
def test_backbones_voxelise_elastic_coordinates_like_the_oracle(backbone):
    """forward_wrapper with targets['elastic_coords'] (minkunet.py:606-608, 665-682; spconvunet.py:291-294, 337-352):
    features, distorted superpoint positions and undistorted positions against the oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _det import det_param
    from oracle import sparse_ref as R
    d = dev()
    if backbone == "mink":
        from segdino3d_amd.backbone_mink import Res16UNet34C
        pts, tgt = _elastic_scene(20000, 150, 14)
        m = Res16UNet34C(in_channels=259, out_channels=96, config=dict(dilations=[1, 1, 1, 1], conv1_kernel_size=5, bn_momentum=0.02),
                         voxel_size=0.02, mode_fuse_2d_feat="early_fusion", add_positional_embedding=True).eval()
    else:
        from segdino3d_amd.backbone_spconv import SpConvUNet
        pts, tgt = _elastic_scene(10000, 100, 15)
        m = SpConvUNet(num_planes=[32 * (i + 1) for i in range(5)], return_blocks=True, voxel_size=0.02,
                       mode_fuse_2d_feat="early_fusion", add_positional_embedding=True).eval()
    sd = {k: det_param("backbone." + k, v.shape).to(v.dtype) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.to(d)
    f, pos, pos_wo = m.forward_wrapper([pts], [tgt], return_sp_mean_pos=True)
    ref_sd = {"backbone." + k: v for k, v in sd.items()}
    fn = R.mink_forward_wrapper if backbone == "mink" else R.spconv_forward_wrapper
    ef = tgt["extra_features"]
    rf, rp, rp_wo = fn(ref_sd, pts.cpu(), ef["points_2dfeats"].cpu(), ef["super_point_masks"].cpu(), elastic=tgt["elastic_coords"].cpu())
    torch.testing.assert_close(pos[0].cpu(), rp, rtol=5e-5, atol=1e-4)
    torch.testing.assert_close(pos_wo[0].cpu(), rp_wo, rtol=5e-5, atol=1e-4)
    assert (pos[0].cpu() - pos_wo[0].cpu()).abs().max().item() > 0.1           # the distortion is visible in the positions
    err = (f[0].cpu() - rf).abs().max().item()
    assert err <= 2e-3 * max(rf.abs().max().item(), 1.0), f"{backbone}: features differ by {err}"
