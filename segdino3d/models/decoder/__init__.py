from segdino3d_amd.decoder import ScanNetQueryDecoder  # noqa: F401

__all__ = ["ScanNetQueryDecoder"]
