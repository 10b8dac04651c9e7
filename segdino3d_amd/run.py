"""`python -m segdino3d_amd.run <script.py> [args...]`: run one of the reference's scripts (train_3d.py, ...) unchanged on the
MI355X path.  A script's own directory precedes PYTHONPATH on sys.path, so `import segdino3d` inside `train_3d.py` would find
the reference's package (and its MinkowskiEngine / spconv imports); this launcher installs the alias package first
(segdino3d_amd/install.py) and then executes the script as `__main__` exactly like `python script.py` would.

    cd /path/to/SegDINO3D
    PYTHONPATH=/path/to/this/repo python -m torch.distributed.run --nproc-per-node 8 -m segdino3d_amd.run train_3d.py --config_file ...
"""
from __future__ import annotations

import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m segdino3d_amd.run <script.py> [args...]")
    script = os.path.abspath(argv[0])
    from .install import install
    install(os.environ.get("SEGDINO3D_REFERENCE_ROOT") or os.path.dirname(script))
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))           # what `python script.py` does
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
