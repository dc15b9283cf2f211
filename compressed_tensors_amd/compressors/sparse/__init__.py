from .marlin_24 import Marlin24Compressor
from .sparse_24_bitmask import (
    Sparse24BitMaskCompressor,
    Sparse24BitMaskTensor,
    get_24_bytemasks,
    sparse24_bitmask_compress,
    sparse24_bitmask_decompress,
)
from .sparse_bitmask import BitmaskCompressor, BitmaskTensor, bitmask_compress, bitmask_decompress

__all__ = [
    "BitmaskCompressor",
    "BitmaskTensor",
    "bitmask_compress",
    "bitmask_decompress",
    "Sparse24BitMaskCompressor",
    "Sparse24BitMaskTensor",
    "sparse24_bitmask_compress",
    "sparse24_bitmask_decompress",
    "get_24_bytemasks",
    "Marlin24Compressor",
]
