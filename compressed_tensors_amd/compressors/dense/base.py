"""Identity codec (reference compressors/dense/base.py:12-60)."""
from ...config import CompressionFormat
from ..base import BaseCompressor

__all__ = ["DenseCompressor"]


@BaseCompressor.register(name=CompressionFormat.dense.value)
class DenseCompressor(BaseCompressor):
    @classmethod
    def compression_param_names(cls, scheme) -> tuple:
        return ("weight",)

    @classmethod
    def compress(cls, state_dict: dict, scheme) -> dict:
        return state_dict

    @classmethod
    def decompress(cls, state_dict: dict, scheme) -> dict:
        return state_dict

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        return True
