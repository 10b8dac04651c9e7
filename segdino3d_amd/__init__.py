"""MI355X-native SegDINO3D forward path.  Host-side mirror of the reference's
`segdino3d` package surface for the hot path; the arithmetic lives in csrc/ (HIP, gfx950) behind
the C ABI declared in include/segdino3d_hip.h."""
from .builder import *  # noqa: F401,F403
from .gtypes import GDType, GD3DTarget  # noqa: F401
