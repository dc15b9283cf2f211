"""compressed_tensors_amd — the compress/decompress hot path of compressed-tensors, rebuilt for
AMD MI355X (gfx950): hand-written HIP kernels behind the reference's Compressor /
ModelCompressor plug-in surface.  See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"

from . import codec
from .compressors import (
    BaseCompressor,
    BitmaskCompressor,
    DenseCompressor,
    FloatQuantizationCompressor,
    IntQuantizationCompressor,
    Marlin24Compressor,
    ModelCompressor,
    MXFP4PackedCompressor,
    MXFP8QuantizationCompressor,
    NVFP4PackedCompressor,
    NaiveQuantizationCompressor,
    PackedQuantizationCompressor,
    Sparse24BitMaskCompressor,
    compress_module,
    decompress_module,
)
from .config import CompressionFormat, SparsityStructure
from .quantization import (
    QuantizationArgs,
    QuantizationScheme,
    QuantizationStatus,
    QuantizationStrategy,
    QuantizationType,
    dequantize,
    fake_quantize,
    quantize,
)

__all__ = [
    "codec",
    "BaseCompressor",
    "ModelCompressor",
    "compress_module",
    "decompress_module",
    "DenseCompressor",
    "NaiveQuantizationCompressor",
    "IntQuantizationCompressor",
    "FloatQuantizationCompressor",
    "PackedQuantizationCompressor",
    "BitmaskCompressor",
    "Sparse24BitMaskCompressor",
    "Marlin24Compressor",
    "NVFP4PackedCompressor",
    "MXFP4PackedCompressor",
    "MXFP8QuantizationCompressor",
    "CompressionFormat",
    "SparsityStructure",
    "QuantizationArgs",
    "QuantizationScheme",
    "QuantizationStatus",
    "QuantizationStrategy",
    "QuantizationType",
    "quantize",
    "dequantize",
    "fake_quantize",
]
