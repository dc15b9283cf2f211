from .base import MXFP4PackedCompressor, NVFP4PackedCompressor

__all__ = ["NVFP4PackedCompressor", "MXFP4PackedCompressor"]
