"""`segdino3d.gtypes` (reference `segdino3d/gtypes.py`): the target container the datasets fill and the model reads."""
from segdino3d_amd.gtypes import *  # noqa: F401,F403
from segdino3d_amd.gtypes import GD3DTarget, GDType  # noqa: F401
