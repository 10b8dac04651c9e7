from segdino3d_amd.backbone_mink import Res16UNet34C  # noqa: F401
from segdino3d_amd.backbone_spconv import SpConvUNet  # noqa: F401

__all__ = ["Res16UNet34C", "SpConvUNet"]
