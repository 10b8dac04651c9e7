"""MI355X-native SegDINO3D forward path.

Host-side mirror of the reference's `segdino3d` package surface for the hot path: the registries and
`build_*` helpers of `segdino3d/builder.py`, the target container of `segdino3d/gtypes.py`, and the
registered classes `Baseline3D`, `Res16UNet34C`, `SpConvUNet`, `ScanNetQueryDecoder`.  The arithmetic
lives in csrc/ (HIP, gfx950) behind the C ABI declared in include/segdino3d_hip.h and is loaded lazily
on first use; importing this package never needs a GPU.
"""
from .builder import *  # noqa: F401,F403
from .builder import __all__ as _builder_all
from .gtypes import GDType, GD3DTarget  # noqa: F401
from .backbone_mink import Res16UNet34C  # noqa: F401
from .backbone_spconv import SpConvUNet  # noqa: F401
from .decoder import ScanNetQueryDecoder  # noqa: F401
from .architecture import Baseline3D, PointData  # noqa: F401
from ._trace import capture  # noqa: F401

__all__ = list(_builder_all) + ["GDType", "GD3DTarget", "Res16UNet34C", "SpConvUNet", "ScanNetQueryDecoder", "Baseline3D", "PointData", "capture"]
