"""naive-quantized / int-quantized / float-quantized codecs
(reference compressors/naive_quantized/base.py:27-164).  INT weights are quantized to int8 by
the HIP quantize kernel and dequantized by the HIP dequantize kernel; FLOAT 8-bit weights become
float8_e4m3fn through the same kernels (clamp to +-448, v_cvt_pk_fp8_f32)."""
from ... import codec
from ...config import CompressionFormat
from ...quantization.quant_args import enum_value
from ...utils import getattr_chain
from ..base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor

__all__ = ["NaiveQuantizationCompressor", "IntQuantizationCompressor", "FloatQuantizationCompressor"]


@BaseCompressor.register(name=CompressionFormat.naive_quantized.value)
class NaiveQuantizationCompressor(BaseCompressor):
    @classmethod
    def compression_param_names(cls, scheme) -> tuple:
        names = ("weight", "weight_scale")
        if not getattr_chain(scheme, "weights.symmetric", True):
            names += ("weight_zero_point",)
        if enum_value(getattr_chain(scheme, "weights.actorder", None)) == "group":
            names += ("weight_g_idx",)
        return names

    @classmethod
    def compress(cls, state_dict: dict, scheme) -> dict:
        """naive_quantized/base.py:48-100.  The block-strategy pad/truncate of the reference is a
        no-op here: the kernel addresses scales per (row // bh, col // bw) without padding."""
        state_dict = state_dict.copy()
        weight = state_dict.pop("weight")
        scale = state_dict.get("weight_scale")
        zero_point = state_dict.get("weight_zero_point", None)
        g_idx = state_dict.get("weight_g_idx", None)
        weights = scheme.weights
        state_dict["weight"] = codec.quantize_tensor(
            weight, scale, zero_point, qtype=enum_value(getattr(weights, "type", "int")),
            num_bits=int(weights.num_bits), strategy=enum_value(weights.strategy),
            group_size=getattr(weights, "group_size", None), block_structure=getattr(weights, "block_structure", None),
            dtype=weights.pytorch_dtype(), g_idx=g_idx,
        )
        return cls._remove_symmetric_zp(state_dict, scheme)

    @classmethod
    def compress_rtn(cls, weight, scheme) -> dict:
        """Round-to-nearest compression straight from the dense weight (min-max qparams): what `compress` returns for
        {"weight", "weight_scale", "weight_zero_point"} of calculate_qparams over the weight's min / max.  Channel-wise
        8-bit schemes (W8A8 int8 / FP8) take ONE pass over the weight (codec.rtn_quantize_channel8); the rest composes the
        observer kernel with `compress`."""
        import torch

        from ...quantization.utils import calculate_qparams_from_weight

        weights = scheme.weights
        qtype = enum_value(getattr(weights, "type", "int"))
        one_pass = (enum_value(weights.strategy) == "channel" and int(weights.num_bits) == 8 and weight.dim() == 2
                    and weight.dtype in (torch.bfloat16, torch.float16) and weight.shape[1] % 8 == 0 and weight.shape[1] <= 16384
                    and (qtype == "int" or (weights.symmetric and getattr(weights, "scale_dtype", None) is None)))
        if not one_pass:
            scale, zp = calculate_qparams_from_weight(weight, weights)
            return cls.compress({"weight": weight, "weight_scale": scale, "weight_zero_point": zp}, scheme)
        q, scale, zp = codec.rtn_quantize_channel8(weight, qtype=qtype, symmetric=bool(weights.symmetric))
        return cls._remove_symmetric_zp({"weight": q, "weight_scale": scale, "weight_zero_point": zp}, scheme)

    @classmethod
    def decompress(cls, state_dict: dict, scheme) -> dict:
        """naive_quantized/base.py:102-126"""
        state_dict = state_dict.copy()
        weight = state_dict.pop("weight")
        scale = state_dict.get("weight_scale")
        zero_point = state_dict.get("weight_zero_point", None)
        g_idx = state_dict.get("weight_g_idx", None)
        state_dict["weight"] = codec.dequantize_tensor(weight, scale, zero_point, g_idx=g_idx)
        return state_dict

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        return module_type in COMPRESSIBLE_MODULE_TYPES and getattr(scheme, "weights", None) is not None


@BaseCompressor.register(name=CompressionFormat.int_quantized.value)
class IntQuantizationCompressor(NaiveQuantizationCompressor):
    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        w = getattr(scheme, "weights", None)
        return (
            module_type in COMPRESSIBLE_MODULE_TYPES
            and getattr(scheme, "input_activations", None) is not None
            and w is not None
            and enum_value(w.type) == "int"
        )


@BaseCompressor.register(name=CompressionFormat.float_quantized.value)
class FloatQuantizationCompressor(NaiveQuantizationCompressor):
    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        w = getattr(scheme, "weights", None)
        return (
            module_type in COMPRESSIBLE_MODULE_TYPES
            and getattr(scheme, "input_activations", None) is not None
            and w is not None
            and enum_value(w.type) == "float"
        )
