"""Drop-in alias: `import segdino3d` resolves to the MI355X-native package, so the reference's
unchanged callers (`train_3d.py:18,141`, `evaluation/evaluator_3d.py`) build the model through the
same names: `from segdino3d import build_architecture`, `ARCHITECTURES`, ...  (SURVEY.md 8(b))."""
from segdino3d_amd import *  # noqa: F401,F403
from segdino3d_amd import __all__  # noqa: F401
from segdino3d_amd import builder, gtypes  # noqa: F401
