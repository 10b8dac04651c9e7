"""Layer-by-layer comparison of the HIP SpConvUNet against the oracle (debug aid, GPU box only)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _det import det_param
from helpers import device_level_coords, match_rows
from oracle import sparse_ref as R
from segdino3d_amd import ops
from segdino3d_amd.backbone_spconv import SpConvUNet
from segdino3d_amd.sparse import SceneMaps
from segdino3d_amd.synth import make_scene

d = torch.device("cuda:0")
pts, tgt = make_scene(15, n_points=10000, n_superpoints=100, n_query2d=20)
m = SpConvUNet(num_planes=[32 * (i + 1) for i in range(5)], return_blocks=True).eval()
sd = {k: det_param("backbone." + k, v.shape).to(v.dtype) for k, v in m.state_dict().items()}
m.load_state_dict(sd); m.to(d)
pk = m.packed()
ef = tgt.extra_features
P = pts.to(d)
maps = SceneMaps(P, 0.02, 5, shift_to_min=True, order="z_fastest", superpoints=ef["super_point_masks"].to(d), clip_min_shape=128)
xyz = pts[:, :3]
c = R.floor_voxel(xyz - xyz.min(0)[0], 0.02)
print("icoords equal:", np.array_equal(maps.icoords.cpu().numpy(), c))
uc, inv = R.unique_voxels(c)
lv = R.SpLevels(uc, 5)
# level coords: device gives absolute coords in stride-1 units; oracle SpLevels coords are in level units
def dev_coords(l):
    return device_level_coords(maps, l) >> l
perm = [match_rows(dev_coords(l), lv.coords[l].astype(np.int64)) for l in range(5)]
print("levels", maps.n_vox, [len(x) for x in lv.coords])
f = torch.cat([pts[:, 3:], xyz - xyz.mean(0), ef["points_2dfeats"]], 1)
vf_ref = R.segment_mean(f, inv, len(uc))
vf = maps.voxel_features(P, ef["points_2dfeats"].to(d), 2, 288)
print("voxel feats err", (vf.cpu()[:, :262] - vf_ref[perm[0]]).abs().max().item())
S = {k[len("backbone."):]: v for k, v in {"backbone." + k: v for k, v in sd.items()}.items()}
x_ref = R.sparse_conv(vf_ref, lv.same(0), R._spw(S["input_conv.0.weight"]), len(uc))
for nt in (0, 1, -11):
    x = ops.gather_gemm(vf, pk["input_conv.0"], nbr=maps.same(0, 3), nt=nt)
    print(f"input_conv nt={nt} err", (x.cpu() - x_ref[perm[0]]).abs().max().item(), "scale", x_ref.abs().max().item())
x = ops.gather_gemm(vf, pk["input_conv.0"], nbr=maps.same(0, 3))
# block0 at level 0
b_ref = R._sp_resblock(x_ref, S, "blocks.block0", lv.same(0))
b = m._resblock(pk, "blocks.block0", x, maps.same(0, 3))
print("block0 err", (b.cpu() - b_ref[perm[0]]).abs().max().item(), "scale", b_ref.abs().max().item())
b1_ref = R._sp_resblock(b_ref, S, "blocks.block1", lv.same(0))
b1 = m._resblock(pk, "blocks.block1", b, maps.same(0, 3))
print("block1 err", (b1.cpu() - b1_ref[perm[0]]).abs().max().item())
h_ref = torch.relu(R.bn_eval(b1_ref, S, "conv.0", R.SPCONV_EPS))
h_ref = R.sparse_conv(h_ref, lv.pairs_down[0], R._spw(S["conv.2.weight"]), len(lv.coords[1]))
s_, b_ = pk["conv.0"]
h = ops.gather_gemm(ops.scale_shift_act(b1, s_, b_, act="relu"), pk["conv.2"], nbr=maps.down(0))
print("down conv err", (h.cpu() - h_ref[perm[1]]).abs().max().item(), "scale", h_ref.abs().max().item())
full_ref = R._sp_unet(x_ref, S, "", lv, 0, 5)
full = m._unet(pk, "", maps, 0, x)
e = (full.cpu() - full_ref[perm[0]]).abs()
print("unet err", e.max().item(), "scale", full_ref.abs().max().item(), "rows>1e-4:", int((e.amax(1) > 1e-4).sum()), "of", e.shape[0])
