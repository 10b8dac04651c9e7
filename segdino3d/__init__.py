"""Drop-in alias: `import segdino3d` resolves to the MI355X-native model side and falls through to the reference checkout
for the pure-Python rest (`segdino3d.utils`, `segdino3d.datasets.*`, `segdino3d.models.module`).  How and why:
segdino3d_amd/install.py.  Mirrors the export list of the reference's `segdino3d/__init__.py:6-47`."""
from segdino3d_amd import *  # noqa: F401,F403
from segdino3d_amd import __all__ as _amd_all
from segdino3d_amd.install import fallthrough_paths as _fallthrough

__path__ = list(__path__) + [p for p in _fallthrough("") if p not in __path__]      # noqa: F821 - set by the import system

from . import builder, gtypes, models  # noqa: E402,F401
from .models.architecture import *  # noqa: E402,F401,F403
from .models.backbone import *  # noqa: E402,F401,F403
from .models.decoder import *  # noqa: E402,F401,F403
from .models.loss import *  # noqa: E402,F401,F403

__all__ = list(_amd_all)

# The reference registers its datasets / preparers / transforms when the package is imported (`__init__.py:21-23`) and its
# unchanged callers rely on that (`utils/dataset_utils.py:6` builds them by name).  They are the reference's own files, found
# through the fall-through path; their third-party imports (mmdet3d, torch_scatter, PIL, ...) are the host's business, so a
# failure here is kept for the caller to inspect instead of breaking model-only use.
datasets_import_error = None
if len(__path__) > 1:
    try:
        from .datasets.dataset import *  # noqa: E402,F401,F403
        from .datasets.preparer import *  # noqa: E402,F401,F403
        from .datasets.transform import *  # noqa: E402,F401,F403
    except Exception as _e:  # noqa: BLE001  # pragma: no cover - depends on the host's Python environment (any failure of the
        # reference's dataset side - missing wheels, registry clashes, version mismatches - must not break model-only use)
        datasets_import_error = _e
