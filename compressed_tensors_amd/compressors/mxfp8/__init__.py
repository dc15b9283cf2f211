from .base import MXFP8QuantizationCompressor

__all__ = ["MXFP8QuantizationCompressor"]
