"""`segdino3d.builder` (reference `segdino3d/builder.py:1-82`): the registries and `build_*` helpers of the AMD package."""
from segdino3d_amd.builder import *  # noqa: F401,F403
from segdino3d_amd.builder import __all__  # noqa: F401
