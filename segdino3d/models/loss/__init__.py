from segdino3d_amd.criterion import ScanNetUnifiedCriterion  # noqa: F401

__all__ = ["ScanNetUnifiedCriterion"]
