"""marlin-24 codec: 2:4 semi-structured sparsity + int4/int8 weights in the Marlin-24 layout.

The format id survives in the reference (config/base.py:23) with its building blocks
(utils/semi_structured_conversions.py:66-197, utils/permutations_24.py:20-53); the compressor
class was removed.  Restated pipeline (compress only, as upstream):

    W, scale -> fp16; q = quantize(W, scale, zp, args) kept in fp16
    (q_comp, meta) = cutlass 2:4 compress(q)            # zeros detected before the unsigned shift
    weight_packed = marlin24_pack((q_comp.T + 2^(b-1)))  # 16x16 tile permutation, 32/b codes per int32
    scale_packed  = scale.T permuted (scale_perm for group, scale_perm_single for channel)
    meta          = meta viewed as (meta_cols / 2, rows * 2)

All device work is HIP: ONE fused front-end kernel (fp16 quantize + 2:4 structure check + 2:4
compress: `ct_marlin24_quant_compress`, no full-size intermediate), one packing kernel that reads
the un-transposed int8 codes (the transpose is index arithmetic), one scale kernel.
"""
import torch

from ... import codec
from ...config import CompressionFormat
from ...quantization.quant_args import enum_value
from ...utils.helpers import tensor_follows_mask_structure
from ..base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor

__all__ = ["Marlin24Compressor"]


@BaseCompressor.register(name=CompressionFormat.marlin_24.value)
class Marlin24Compressor(BaseCompressor):
    COMPRESSION_PARAM_NAMES = ("weight_packed", "scale_packed", "meta")

    @staticmethod
    def validate_quant_compatability(weights) -> bool:
        st = enum_value(weights.strategy)
        if st not in ("group", "channel"):
            raise ValueError(f"Marlin24 Compressor is only valid for group and channel quantization strategies, got {st}")
        if not weights.symmetric:
            raise ValueError("Marlin24 Compressor is only valid for symmetric quantization, got symmetric=False")
        if weights.num_bits not in (4, 8):
            raise ValueError(f"Marlin24 Compressor is only valid for 4 or 8 bit quantization, got {weights.num_bits}")
        return True

    @staticmethod
    def validate_sparsity_structure(name: str, weight: torch.Tensor) -> bool:
        if not tensor_follows_mask_structure(weight):
            raise ValueError("Marlin24 Compressor is only compatible with weights that have a 2:4 sparsity structure. "
                             f"Found segments in {name} that do not match the expected structure.")
        return True

    @classmethod
    def compression_param_names(cls, scheme=None) -> tuple:
        return cls.COMPRESSION_PARAM_NAMES

    @classmethod
    def compress(cls, state_dict: dict, scheme) -> dict:
        state_dict = state_dict.copy()
        weight = state_dict.pop("weight")
        scale = state_dict.pop("weight_scale")
        zero_point = state_dict.pop("weight_zero_point", None)
        weights = scheme.weights
        cls.validate_quant_compatability(weights)

        group_size = getattr(weights, "group_size", None)
        fused_ok = (weight.dtype in (torch.float16, torch.bfloat16) and scale.dtype in (torch.float16, torch.bfloat16) and weight.dim() == 2
                    and weight.shape[0] % 64 == 0 and weight.shape[1] % 16 == 0
                    and (enum_value(weights.strategy) == "channel" or (group_size and group_size % 16 == 0 and weight.shape[1] % group_size == 0)))
        bad = None
        if fused_ok:
            # one pass: weight.to(fp16) / scale.to(fp16) / quantize in fp16 / 2:4 structure check / cutlass 2:4 compress.
            # Everything is queued before the flag is read, so the host work below overlaps the kernels
            g = None if enum_value(weights.strategy) == "channel" else group_size
            size_n, size_k = weight.shape[0], weight.shape[1] // 2
            if int(weights.num_bits) == 4 and weight.shape[1] % 256 == 0:
                # everything in one launch: no int8 intermediate, no separate packing kernel
                packed, meta, bad = codec.marlin24_compress_w4(weight, scale, zero_point, group_size=g)
            else:
                comp, meta, bad = codec.marlin24_quant_compress(weight, scale, zero_point, num_bits=int(weights.num_bits), group_size=g)
                packed = codec.marlin24_pack_weights(comp, int(weights.num_bits), transposed=True, add_offset=True)
        else:
            scale16 = scale.to(torch.float16)
            w16 = weight.to(torch.float16)
            q = codec.quantize_tensor(
                w16, scale16, zero_point, num_bits=int(weights.num_bits), strategy=enum_value(weights.strategy),
                group_size=group_size,
            )
            cls.validate_sparsity_structure("weight", q)
            comp, meta = codec.cutlass24_from_dense(q)
            size_n, size_k = comp.shape  # the kernel expects input-dim first: packed from comp.T
            packed = codec.marlin24_pack_weights(comp, int(weights.num_bits), transposed=True, add_offset=True)
        is_group = enum_value(weights.strategy) == "group" and group_size is not None and group_size < size_k
        scale2d = scale.reshape(scale.shape[0], -1)
        if scale2d.dtype in (torch.float16, torch.bfloat16):
            scale_packed = codec.marlin24_pack_scales(scale2d, single=not is_group, to_float16=True)
        else:
            scale_packed = codec.marlin24_pack_scales(scale2d.to(torch.float16), single=not is_group)
        meta = meta.reshape(-1).reshape(meta.shape[1] // 2, meta.shape[0] * 2)
        if bad is not None and int(bad.item()):  # one host read, as the reference pipeline's structure check
            raise ValueError("Marlin24 Compressor is only compatible with weights that have a 2:4 sparsity structure. "
                             "Found segments in weight that do not match the expected structure.")

        state_dict["weight_packed"] = packed
        state_dict["scale_packed"] = scale_packed
        state_dict["meta"] = meta
        return state_dict

    @classmethod
    def decompress(cls, state_dict: dict, scheme) -> dict:
        raise NotImplementedError("Decompression is not implemented for the Marlin24 Compressor.")

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        return False  # explicit opt-in only, like every sparsity format
