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


@BaseCompressor.register(name=CompressionFormat.mxfp8_quantized.value)
class MXFP8QuantizationCompressor(NaiveQuantizationCompressor):
    @classmethod
    def _compress_scale(cls, scale: torch.Tensor, weights) -> torch.Tensor:
        return codec.compress_mx_scale(scale, getattr(weights, "scale_dtype", None) or torch.uint8)

    @classmethod
    def _decompress_scale(cls, scale: torch.Tensor) -> torch.Tensor:
        return codec.decompress_mx_scale(scale)

    @classmethod
    def compress(cls, state_dict: dict, scheme) -> dict:
        """mxfp8/base.py:47-72: quantize with the float scale, then store the scale as its E8M0 exponent"""
        state_dict = NaiveQuantizationCompressor.compress.__func__(cls, state_dict, scheme)
        state_dict["weight_scale"] = cls._compress_scale(state_dict["weight_scale"], scheme.weights)
        return state_dict

    @classmethod
    def decompress(cls, state_dict: dict, scheme) -> dict:
        """mxfp8/base.py:74-101: the scale comes back as bfloat16, so the weight does too"""
        state_dict = state_dict.copy()
        state_dict["weight_scale"] = cls._decompress_scale(state_dict["weight_scale"])
        return NaiveQuantizationCompressor.decompress.__func__(cls, state_dict, scheme)

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        """mxfp8/base.py:103-118: FP8 with group_size 32 and uint8 scales"""
        w = getattr(scheme, "weights", None)
        return (
            module_type in COMPRESSIBLE_MODULE_TYPES
            and w is not None
            and int(w.num_bits) == 8
            and enum_value(w.type) == "float"
            and getattr(w, "group_size", None) == 32
            and getattr(w, "scale_dtype", None) == torch.uint8
        )
