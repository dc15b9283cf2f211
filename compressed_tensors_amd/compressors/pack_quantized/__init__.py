from .base import PackedQuantizationCompressor
from .helpers import pack_to_int32, unpack_from_int32

__all__ = ["PackedQuantizationCompressor", "pack_to_int32", "unpack_from_int32"]
