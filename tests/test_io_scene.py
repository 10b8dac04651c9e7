"""Input side (segdino3d_amd.io_scene) on the CPU: decoding of the reference's file formats, the val transform
against the reference's own output, the packed format round trip."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _write_reference_files(tmp, scene_id="scene0000_00", N=3000, M=17, seed=0):
    g = torch.Generator().manual_seed(seed)
    for sub in ("points", "instance_mask", "semantic_mask", "super_points"):
        os.makedirs(os.path.join(tmp, "scenes", sub), exist_ok=True)
    os.makedirs(os.path.join(tmp, "feats"), exist_ok=True)
    pts = torch.cat([torch.rand(N, 3, generator=g) * 6, torch.randint(0, 256, (N, 3), generator=g).float()], dim=1)
    pts.numpy().astype(np.float32).tofile(os.path.join(tmp, "scenes", "points", f"{scene_id}.bin"))
    inst = torch.randint(-1, 12, (N,), generator=g)
    sem = torch.randint(0, 200, (N,), generator=g)
    sp = torch.randint(0, 90, (N,), generator=g)
    inst.numpy().astype(np.int64).tofile(os.path.join(tmp, "scenes", "instance_mask", f"{scene_id}.bin"))
    sem.numpy().astype(np.int64).tofile(os.path.join(tmp, "scenes", "semantic_mask", f"{scene_id}.bin"))
    sp.numpy().astype(np.int64).tofile(os.path.join(tmp, "scenes", "super_points", f"{scene_id}.bin"))
    scales = [torch.randn(N, 256, generator=g) for _ in range(3)]
    torch.save(scales, os.path.join(tmp, "feats", f"{scene_id}.pth"))
    qf, qp = torch.randn(M, 256, generator=g), torch.rand(M, 3, generator=g) * 6
    torch.save(qf, os.path.join(tmp, "feats", f"{scene_id}_query_feats.pth"))
    torch.save(qp, os.path.join(tmp, "feats", f"{scene_id}_query_3dctr.pth"))
    return dict(points=pts, inst=inst, sem=sem, sp=sp, scales=scales, qf=qf, qp=qp, scene_id=scene_id)


def test_val_transform_matches_reference_golden():
    from segdino3d_amd import io_scene
    z = np.load(os.path.join(GOLDEN, "val_transform.npz"))
    out = io_scene.normalize_points_color(torch.from_numpy(z["raw"].copy()))
    assert torch.equal(out, torch.from_numpy(z["out"]))                      # same two fp32 operations: bit-exact


def test_reference_files_decode_and_pack_round_trip(tmp_path):
    from segdino3d_amd import io_scene
    ref = _write_reference_files(str(tmp_path))
    sc = io_scene.read_reference_scene(os.path.join(tmp_path, "scenes"), os.path.join(tmp_path, "feats"), ref["scene_id"])
    assert torch.equal(sc["points"][:, :3], ref["points"][:, :3])
    assert torch.equal(sc["points"], io_scene.normalize_points_color(ref["points"].clone()))
    assert torch.equal(sc["points_2dfeats"], torch.stack(ref["scales"], dim=0).mean(dim=0))      # scannet200.py:234-235
    assert torch.equal(sc["super_points"], ref["sp"]) and torch.equal(sc["instance_mask"], ref["inst"])
    assert torch.equal(sc["query2d_feats"], ref["qf"]) and torch.equal(sc["query2d_pos"], ref["qp"])
    path = os.path.join(tmp_path, "scene.sd3d")
    nbytes = io_scene.pack_scene(path, sc)
    assert nbytes == os.path.getsize(path) and nbytes % 256 == 0
    back = io_scene.load_packed(path)
    for k in ("points", "super_points", "points_2dfeats", "query2d_feats", "query2d_pos", "instance_mask", "semantic_mask"):
        assert torch.equal(back[k], sc[k].reshape(back[k].shape)), k
    # half-precision features: smaller file, values rounded to fp16
    path16 = os.path.join(tmp_path, "scene16.sd3d")
    assert io_scene.pack_scene(path16, sc, feats_fp16=True) < nbytes * 0.55
    b16 = io_scene.load_packed(path16)
    assert b16["points_2dfeats"].dtype == torch.float16
    assert torch.equal(b16["points_2dfeats"], sc["points_2dfeats"].half()) and torch.equal(b16["points"], sc["points"])
    with open(path, "r+b") as f:                                              # a corrupted header is rejected
        f.write(b"XXXXXXXX")
    with pytest.raises(ValueError):
        io_scene.load_packed(path)
