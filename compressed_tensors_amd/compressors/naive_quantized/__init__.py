from .base import FloatQuantizationCompressor, IntQuantizationCompressor, NaiveQuantizationCompressor

__all__ = ["NaiveQuantizationCompressor", "IntQuantizationCompressor", "FloatQuantizationCompressor"]
