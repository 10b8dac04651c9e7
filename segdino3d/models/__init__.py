"""`segdino3d.models`: architecture / backbone / decoder / loss are the AMD classes; `segdino3d.models.module` (NestedTensor,
the reference attention module, positional encodings - imported by `utils/dataset_utils.py:7`) falls through to the reference."""
from segdino3d_amd.install import fallthrough_paths as _fallthrough

__path__ = list(__path__) + [p for p in _fallthrough("models") if p not in __path__]   # noqa: F821

from . import architecture, backbone, decoder, loss  # noqa: E402,F401
