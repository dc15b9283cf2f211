from .base import DenseCompressor

__all__ = ["DenseCompressor"]
