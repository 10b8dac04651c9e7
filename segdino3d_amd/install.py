"""Making `import segdino3d` resolve to the MI355X path for the reference's UNCHANGED callers.

The callers do not only use the registries; they also import pure-Python submodules of the reference package:

    train_3d.py:18-19                    from segdino3d import build_architecture
                                         from segdino3d.utils import get_rank, init_distributed_mode, is_main_process
    evaluation/evaluate_3d.py:9,14-15    from segdino3d.utils import ...; from segdino3d.datasets.dataset import ScanNet200InstanceSeg3D
    utils/dataset_utils.py:6-7           from segdino3d import build_dataset; from segdino3d.models.module import NestedTensor, ...
    segdino3d/datasets/dataset/*.py      from segdino3d import DATASETS, build_transform; from segdino3d.gtypes import GD3DTarget

So the alias package (`<this repo>/segdino3d/`) owns the MODEL side - registries / `build_*` (`segdino3d.builder`), the target
container (`segdino3d.gtypes`), `segdino3d.models.{architecture,backbone,decoder,loss}` with the AMD classes - and FALLS
THROUGH to the reference checkout for everything else: its `__path__` (and that of `segdino3d.models`) lists the reference's
directory after its own, so `segdino3d.utils`, `segdino3d.datasets.*`, `segdino3d.models.module` are the reference's files,
executed unchanged, and their `from segdino3d import DATASETS, ...` lines land in the AMD registries.  The reference's own
`segdino3d/__init__.py` (which imports MinkowskiEngine / spconv through `models.backbone`) is never executed.

Two ways in:
  * this repository FIRST on sys.path (`python -m ...`, `python -c`, pytest, notebooks): `import segdino3d` finds the alias;
  * a script run from the reference checkout (`python train_3d.py`, `torch.distributed.launch train_3d.py`): the script's own
    directory precedes PYTHONPATH, so the reference's package would win - use `python -m segdino3d_amd.run train_3d.py ...`
    (or call `segdino3d_amd.install()` before the first `import segdino3d`), which places the alias in `sys.modules`.
The reference checkout is found through `SEGDINO3D_REFERENCE_ROOT`, the `reference_root` argument, or by scanning sys.path and
the working directory for a `segdino3d/utils/dist_utils.py` that is not ours.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys

ALIAS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "segdino3d")


def find_reference_root(hint: str | None = None) -> str | None:
    """Directory that holds the reference's `segdino3d/` package (the SegDINO3D checkout), or None."""
    cands = []
    if hint:
        cands.append(hint)
    if os.environ.get("SEGDINO3D_REFERENCE_ROOT"):
        cands.append(os.environ["SEGDINO3D_REFERENCE_ROOT"])
    cands += [p or os.getcwd() for p in sys.path] + [os.getcwd()]
    for c in cands:
        pkg = os.path.join(os.path.abspath(c), "segdino3d")
        if os.path.samefile(pkg, ALIAS_DIR) if os.path.isdir(pkg) and os.path.isdir(ALIAS_DIR) else False:
            continue
        if os.path.isfile(os.path.join(pkg, "utils", "dist_utils.py")) or os.path.isdir(os.path.join(pkg, "datasets")):
            return os.path.abspath(c)
    return None


def fallthrough_paths(subdir: str = "", reference_root: str | None = None):
    """Extra `__path__` entries for the alias package (`subdir=""`) or one of its sub-packages (`"models"`)."""
    root = find_reference_root(reference_root)
    if root is None:
        return []
    p = os.path.join(root, "segdino3d", subdir) if subdir else os.path.join(root, "segdino3d")
    return [p] if os.path.isdir(p) else []


def install(reference_root: str | None = None, force: bool = True):
    """Put the alias package into `sys.modules['segdino3d']` (idempotent).  Needed only when something else named
    `segdino3d` precedes this repository on sys.path; returns the module."""
    if reference_root:
        os.environ["SEGDINO3D_REFERENCE_ROOT"] = os.path.abspath(reference_root)
    cur = sys.modules.get("segdino3d")
    if cur is not None:
        if os.path.dirname(os.path.abspath(getattr(cur, "__file__", "") or "")) == ALIAS_DIR:
            return cur
        if not force:
            raise RuntimeError(f"another `segdino3d` is already imported from {getattr(cur, '__file__', '?')}")
        for name in [n for n in sys.modules if n == "segdino3d" or n.startswith("segdino3d.")]:
            del sys.modules[name]
    spec = importlib.util.spec_from_file_location("segdino3d", os.path.join(ALIAS_DIR, "__init__.py"),
                                                  submodule_search_locations=[ALIAS_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["segdino3d"] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop("segdino3d", None)
        raise
    return mod
