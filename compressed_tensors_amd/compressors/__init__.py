from .base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor, compress_module, decompress_module
from .dense import DenseCompressor
from .fp4 import MXFP4PackedCompressor, NVFP4PackedCompressor
from .format import infer_model_format, infer_module_format
from .model_compressors import ModelCompressor
from .mxfp8 import MXFP8QuantizationCompressor
from .naive_quantized import FloatQuantizationCompressor, IntQuantizationCompressor, NaiveQuantizationCompressor
from .pack_quantized import PackedQuantizationCompressor, pack_to_int32, unpack_from_int32
from .sparse import (
    BitmaskCompressor,
    BitmaskTensor,
    Marlin24Compressor,
    Sparse24BitMaskCompressor,
    Sparse24BitMaskTensor,
)

__all__ = [
    "BaseCompressor",
    "COMPRESSIBLE_MODULE_TYPES",
    "compress_module",
    "decompress_module",
    "infer_module_format",
    "infer_model_format",
    "ModelCompressor",
    "DenseCompressor",
    "NaiveQuantizationCompressor",
    "IntQuantizationCompressor",
    "FloatQuantizationCompressor",
    "PackedQuantizationCompressor",
    "pack_to_int32",
    "unpack_from_int32",
    "BitmaskCompressor",
    "BitmaskTensor",
    "Sparse24BitMaskCompressor",
    "Sparse24BitMaskTensor",
    "Marlin24Compressor",
    "NVFP4PackedCompressor",
    "MXFP4PackedCompressor",
    "MXFP8QuantizationCompressor",
]
