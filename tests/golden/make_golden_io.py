"""tests/golden/val_transform.npz: the reference's validation transform (`Scannet200Transforms("val")`,
datasets/transform/wrappers_3d.py:6-57 -> NormalizePointsColor, point_cloud_transforms.py:355-389) applied to seeded
raw points (xyz + rgb 0..255).  Build container only; the fixture is data."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

mg.install_stand_ins()
sys.path.insert(0, mg.REFERENCE)
import segdino3d.datasets.transform.wrappers_3d as W  # noqa: E402

g = torch.Generator().manual_seed(11)
pts = torch.cat([torch.rand(4096, 3, generator=g) * 8 - 4, torch.randint(0, 256, (4096, 3), generator=g).float()], dim=1)
out, _ = W.Scannet200Transforms("val")(pts.clone(), {})
np.savez_compressed(os.path.join(HERE, "val_transform.npz"), raw=pts.numpy(), out=out.numpy())
print("wrote val_transform.npz", out.dtype, out.shape, float(out[:, 3:].mean()))
