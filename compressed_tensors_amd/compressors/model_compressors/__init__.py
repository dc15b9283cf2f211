from .model_compressor import ModelCompressor

__all__ = ["ModelCompressor"]
