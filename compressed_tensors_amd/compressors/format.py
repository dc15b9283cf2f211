"""Format inference (reference compressors/format.py:18-27,75-117)."""
from ..config import CompressionFormat
from ..quantization.utils import is_module_quantized

__all__ = ["infer_module_format", "infer_model_format", "COMPRESSION_FORMAT_PRIORITY"]

# more specific formats first (format.py:18-27)
COMPRESSION_FORMAT_PRIORITY = [
    CompressionFormat.mxfp4_pack_quantized,
    CompressionFormat.mxfp8_quantized,
    CompressionFormat.nvfp4_pack_quantized,
    CompressionFormat.int_quantized,
    CompressionFormat.pack_quantized,
    CompressionFormat.float_quantized,
    CompressionFormat.naive_quantized,
    CompressionFormat.dense,
]


def infer_module_format(module_type: type, scheme) -> CompressionFormat:
    from .base import BaseCompressor

    for fmt in COMPRESSION_FORMAT_PRIORITY:
        try:
            comp = BaseCompressor.get_value_from_registry(fmt.value)
        except KeyError:
            continue
        if comp.can_compress(module_type, scheme):
            return fmt
    raise StopIteration(f"no registered compressor can compress {module_type} with {scheme}")


def infer_model_format(model, force_compression_format=None) -> CompressionFormat:
    formats = set()
    for _, module in model.named_modules(remove_duplicate=True):
        if not is_module_quantized(module):
            continue
        scheme = module.quantization_scheme
        fmt = infer_module_format(type(module), scheme)
        if force_compression_format is not None:
            fmt = force_compression_format
        elif getattr(scheme, "format", None) is not None:
            fmt = scheme.format
        fmt = CompressionFormat(getattr(fmt, "value", fmt))
        try:
            scheme.format = fmt
        except Exception:
            scheme.format = fmt.value
        if fmt != CompressionFormat.dense:
            formats.add(fmt)
    if not formats:
        return CompressionFormat.dense
    if len(formats) == 1:
        return next(iter(formats))
    return CompressionFormat.mixed_precision
