from segdino3d_amd.architecture import Baseline3D  # noqa: F401

__all__ = ["Baseline3D"]
