"""mxfp8-quantized codec (reference compressors/mxfp8/base.py:29-118): float8_e4m3fn weights in groups of 32 under
E8M0 (power-of-two) scales stored as uint8 exponents.  The weight path is the FLOAT 8-bit quantize / dequantize
kernels of the naive-quantized codec; only the (1/32-size) scale tensor is converted around them."""
import torch

from ... import codec
from ...config import CompressionFormat
from ...quantization.quant_args import enum_value
from ..base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor
from ..naive_quantized import NaiveQuantizationCompressor

__all__ = ["MXFP8QuantizationCompressor"]

_naive_compress = NaiveQuantizationCompressor.compress.__func__
_naive_decompress = NaiveQuantizationCompressor.decompress.__func__


@BaseCompressor.register(name=CompressionFormat.mxfp8_quantized.value)
class MXFP8QuantizationCompressor(NaiveQuantizationCompressor):
    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        """mxfp8/base.py:103-118: FLOAT, 8 bits, groups of 32, uint8 (E8M0) scales"""
        w = getattr(scheme, "weights", None)
        if module_type not in COMPRESSIBLE_MODULE_TYPES or w is None:
            return False
        signature = (enum_value(w.type), int(w.num_bits), getattr(w, "group_size", None), getattr(w, "scale_dtype", None))
        return signature == ("float", 8, 32, torch.uint8)

    @classmethod
    def compress(cls, state_dict: dict, scheme) -> dict:
        """mxfp8/base.py:47-72: quantize against the float scale, then keep only the scale's exponent"""
        out = _naive_compress(cls, state_dict, scheme)
        out["weight_scale"] = cls._compress_scale(out["weight_scale"], scheme.weights)
        return out

    @classmethod
    def decompress(cls, state_dict: dict, scheme) -> dict:
        """mxfp8/base.py:74-101: the exponent comes back as a bfloat16 power of two, so the weight is bfloat16 too"""
        widened = dict(state_dict, weight_scale=cls._decompress_scale(state_dict["weight_scale"]))
        return _naive_decompress(cls, widened, scheme)

    # the two hooks keep upstream's names: install() lets upstream's subclass call them
    @classmethod
    def _compress_scale(cls, scale: torch.Tensor, weights) -> torch.Tensor:
        return codec.compress_mx_scale(scale, getattr(weights, "scale_dtype", None) or torch.uint8)

    @classmethod
    def _decompress_scale(cls, scale: torch.Tensor) -> torch.Tensor:
        return codec.decompress_mx_scale(scale)
