"""pack-quantized codec (reference compressors/pack_quantized/base.py:35-177): INT weights of
1..8 bits packed densely into int32 words.

MI355X design: compress is ONE fused kernel (quantize -> clamp/round -> bitstream pack) and
decompress is ONE fused kernel (unpack -> dequantize); the int8 intermediate of the reference
never exists.  Algorithmic traffic per direction at W4A16 g128: 2 B/elem weight + 0.5 B/elem
packed + scales.
"""
import math

import torch

from ... import codec
from ...config import CompressionFormat
from ...quantization.quant_args import enum_value
from ...utils import getattr_chain
from ..base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor

__all__ = ["PackedQuantizationCompressor"]

PACK_ZP_STRATS = ("group", "channel")


def _layout_kwargs(weights):
    return dict(
        num_bits=int(weights.num_bits),
        strategy=enum_value(weights.strategy),
        group_size=getattr(weights, "group_size", None),
        block_structure=getattr(weights, "block_structure", None),
    )


@BaseCompressor.register(name=CompressionFormat.pack_quantized.value)
class PackedQuantizationCompressor(BaseCompressor):
    @classmethod
    def compression_param_names(cls, scheme) -> tuple:
        """base.py:44-60"""
        names = ("weight_packed", "weight_scale", "weight_shape")
        if not getattr_chain(scheme, "weights.symmetric", True):
            names += ("weight_zero_point",)
        if enum_value(getattr_chain(scheme, "weights.actorder", None)) == "group":
            names += ("weight_g_idx",)
        if enum_value(getattr_chain(scheme, "input_activations.strategy", None)) == "tensor_group":
            names += ("input_global_scale",)
        return names

    @classmethod
    def compress(cls, state_dict: dict, scheme, _prepacked=None, _prezp=None) -> dict:
        """base.py:62-114"""
        state_dict = state_dict.copy()
        weight = state_dict.pop("weight")
        scale = state_dict.get("weight_scale")
        zero_point = state_dict.get("weight_zero_point", None)
        g_idx = state_dict.get("weight_g_idx", None)
        weights = scheme.weights

        if weight.device.type == "meta":
            packed_cols = math.ceil(weight.shape[-1] * weights.num_bits / 32)
            state_dict["weight_packed"] = torch.empty((*weight.shape[:-1], packed_cols), dtype=torch.int32, device="meta")
            state_dict["weight_shape"] = torch.tensor(weight.shape)
            return cls._remove_symmetric_zp(state_dict, scheme)

        if enum_value(getattr(weights, "type", "int")) != "int":
            raise NotImplementedError("pack-quantized requires INT weights")
        if _prepacked is not None:
            state_dict["weight_packed"] = _prepacked  # produced by the batched launch of compress_modules
        else:
            state_dict["weight_packed"] = codec.quantize_and_pack(weight, scale, zero_point, g_idx=g_idx, **_layout_kwargs(weights))
        state_dict["weight_shape"] = torch.tensor(weight.shape)  # int64, CPU: as upstream (:105)

        if not weights.symmetric and enum_value(weights.strategy) in PACK_ZP_STRATS:
            assert zero_point is not None, "Asymmetric quant requires zero-point values"
            if _prezp is not None:
                state_dict["weight_zero_point"] = _prezp  # packed by the batched launch of compress_modules
            else:
                zp8 = zero_point if zero_point.dtype is torch.int8 else zero_point.to(torch.int8)
                state_dict["weight_zero_point"] = codec.pack_to_int32(zp8, weights.num_bits, packed_dim=0)

        return cls._remove_symmetric_zp(state_dict, scheme)

    @classmethod
    def decompress(cls, state_dict: dict, scheme, _preweight=None, _prezp=None) -> dict:
        """base.py:116-163"""
        state_dict = state_dict.copy()
        packed = state_dict.pop("weight_packed")
        scale = state_dict.get("weight_scale")
        zero_point = state_dict.get("weight_zero_point", None)
        g_idx = state_dict.get("weight_g_idx", None)
        original_shape = state_dict.get("weight_shape")
        weights = scheme.weights
        shape = tuple(int(s) for s in original_shape.tolist())

        if packed.device.type == "meta":
            state_dict["weight"] = torch.empty(shape, dtype=scale.dtype, device="meta")
            return state_dict

        if not weights.symmetric and enum_value(weights.strategy) in PACK_ZP_STRATS:
            assert zero_point is not None, "Asymmetric quant requires zero-point values"
            zp_shape = (*shape[:-1], scale.shape[-1])
            zero_point = _prezp if _prezp is not None else codec.unpack_from_int32(zero_point, weights.num_bits, zp_shape, packed_dim=0)
            state_dict["weight_zero_point"] = zero_point

        # dequantize() is called without args upstream: the strategy is inferred from the scale
        # shape (base.py:156-161, lifecycle/forward.py:99-130) and group_size from g_idx-free shapes
        if _preweight is not None:
            state_dict["weight"] = _preweight  # produced by the batched launch of decompress_modules
        else:
            state_dict["weight"] = codec.unpack_and_dequantize(
                packed, shape, scale, zero_point, num_bits=int(weights.num_bits), g_idx=g_idx
            )
        return state_dict

    @classmethod
    def compress_rtn(cls, weight: torch.Tensor, scheme) -> dict:
        """Round-to-nearest compression straight from the dense weight (min-max qparams): the state dict
        `compress` returns for {"weight", "weight_scale", "weight_zero_point"} with `calculate_qparams` of the weight's
        group min / max — one pass over the weight for int4 group / channel schemes (codec.rtn_quantize_and_pack), the
        two-kernel composition otherwise."""
        weights = scheme.weights
        strategy = enum_value(weights.strategy)
        group = getattr(weights, "group_size", None) if strategy == "group" else None
        one_pass = (int(weights.num_bits) == 4 and enum_value(getattr(weights, "type", "int")) == "int" and strategy in ("group", "channel")
                    and weight.dim() == 2 and weight.dtype in (torch.bfloat16, torch.float16))
        cols = weight.shape[-1]
        g = int(group) if group else cols
        one_pass = one_pass and g > 0 and cols % g == 0 and g % 32 == 0 and g <= 2048 and ((g // 32) & (g // 32 - 1)) == 0
        if not one_pass:
            scale, zp = codec.minmax_qparams(weight, num_bits=int(weights.num_bits), group_size=group, symmetric=bool(weights.symmetric))
            return cls.compress({"weight": weight, "weight_scale": scale, "weight_zero_point": zp}, scheme)
        packed, scale, zp = codec.rtn_quantize_and_pack(weight, group_size=group, symmetric=bool(weights.symmetric))
        out = {"weight_packed": packed, "weight_scale": scale, "weight_shape": torch.tensor(weight.shape)}
        if not weights.symmetric:
            out["weight_zero_point"] = codec.pack_to_int32(zp, weights.num_bits, packed_dim=0)
        return out

    # ------------------------------------------------------------------ batched module paths
    @classmethod
    def compress_modules(cls, modules) -> None:
        """One launch for every eligible module (int4, 16-bit, group / channel scales), the rest
        one by one.  State-dict results are identical to compress_module's."""
        from ...quantization.quant_args import QuantizationStatus
        from ...utils.module import get_direct_state_dict, replace_direct_state_dict

        batches = {}  # (device, dtype) -> (entries, jobs): one table and one launch per GPU and weight dtype
        for m in modules:
            scheme = m.quantization_scheme
            sd = get_direct_state_dict(m)
            w, scale, zp = sd.get("weight"), sd.get("weight_scale"), sd.get("weight_zero_point")
            wa = scheme.weights
            ok = (w is not None and w.is_cuda and w.is_contiguous() and w.data_ptr() % 16 == 0
                  and enum_value(getattr(wa, "type", "int")) == "int"
                  and codec.w4_batch_eligible(w.shape, w.dtype, scale, zp, num_bits=int(wa.num_bits), strategy=wa.strategy,
                                              group_size=getattr(wa, "group_size", None), g_idx=sd.get("weight_g_idx"), device=w.device))
            if not ok:
                cls.compress_module(m)
                continue
            rows, cols = w.shape
            group = cols if enum_value(wa.strategy) == "channel" else int(wa.group_size)
            packed = torch.empty((rows, cols // 8), dtype=torch.int32, device=w.device)
            entries, jobs = batches.setdefault((w.device, w.dtype), ([], []))
            entries.append((w, scale, zp, packed, rows, cols, group))
            jobs.append((m, sd, scheme, packed))
        for (device, dtype), (entries, jobs) in batches.items():
            codec.W4Batch(entries, "compress", dtype).launch()
            # the zero points of the asymmetric modules: one more launch for all of them (pack_to_int32(zp, 4, packed_dim=0))
            zps = {}
            for i, (m, sd, scheme, packed) in enumerate(jobs):
                zp = sd.get("weight_zero_point")
                if not scheme.weights.symmetric and enum_value(scheme.weights.strategy) in PACK_ZP_STRATS and zp is not None and zp.dtype is torch.int8 \
                        and zp.dim() == 2 and zp.is_contiguous():
                    zps[i] = (zp, torch.empty((math.ceil(zp.shape[0] * 4 / 32), zp.shape[1]), dtype=torch.int32, device=device))
            codec.zp4_batch(zps.values(), "pack")
            for i, (m, sd, scheme, packed) in enumerate(jobs):
                replace_direct_state_dict(m, cls.compress(sd, scheme, _prepacked=packed, _prezp=zps[i][1] if i in zps else None))
                m.quantization_status = QuantizationStatus.COMPRESSED

    @classmethod
    def _batch_decompress(cls, state_dicts, schemes):
        """(weight, unpacked zero point or None) of every eligible state dict from ONE launch (+ one for the packed zero points of the
        asymmetric ones); (None, None) for the others"""
        outs, batches = [(None, None)] * len(state_dicts), {}
        for i, (sd, scheme) in enumerate(zip(state_dicts, schemes)):
            packed, scale, shape_t = sd.get("weight_packed"), sd.get("weight_scale"), sd.get("weight_shape")
            zp_packed = sd.get("weight_zero_point")
            wa = scheme.weights
            # anything unusual (a wrongly typed weight_packed, which must raise as upstream; a zero point kept next to a
            # symmetric scheme, which the per-module path applies; misaligned views) goes through `decompress`
            ok = (packed is not None and scale is not None and shape_t is not None and packed.is_cuda and packed.is_contiguous()
                  and packed.dtype == torch.int32 and packed.data_ptr() % 16 == 0 and sd.get("weight_g_idx") is None)
            if not ok:
                continue
            shape = tuple(int(v) for v in shape_t.tolist())
            if len(shape) != 2 or scale.ndim != 2:
                continue
            # decompress infers the strategy from the scale shape (forward.py:99-130): (R, 1) channel, (R, G) group
            strategy, group = ("channel", shape[-1]) if scale.shape[-1] == 1 else ("group", shape[-1] // scale.shape[-1])
            asym = not wa.symmetric and enum_value(wa.strategy) in PACK_ZP_STRATS
            zp = None
            if asym:
                want = (math.ceil(shape[0] * 4 / 32), scale.shape[-1])
                if not (zp_packed is not None and zp_packed.dtype == torch.int32 and tuple(zp_packed.shape) == want and zp_packed.is_contiguous()
                        and zp_packed.device == packed.device and int(wa.num_bits) == 4):
                    continue
                zp = torch.empty((shape[0], scale.shape[-1]), dtype=torch.int8, device=packed.device)  # filled by the batched unpack below
            elif zp_packed is not None:
                continue
            if not (tuple(packed.shape) == (shape[0], shape[1] // 8)
                    and codec.w4_batch_eligible(shape, scale.dtype, scale, zp, num_bits=int(wa.num_bits), strategy=strategy,
                                                group_size=group, device=packed.device)):
                continue
            out = torch.empty(shape, dtype=scale.dtype, device=packed.device)
            entries, slots, zps = batches.setdefault((packed.device, scale.dtype), ([], [], []))
            entries.append((packed, scale, zp, out, shape[0], shape[1], group))
            slots.append((i, out, zp))
            if asym:
                zps.append((zp_packed, zp))
        for (_, dtype), (entries, slots, zps) in batches.items():
            codec.zp4_batch(zps, "unpack")
            codec.W4Batch(entries, "decompress", dtype).launch()
            for i, out, zp in slots:
                outs[i] = (out, zp)
        return outs

    @classmethod
    def decompress_many(cls, state_dicts, scheme) -> list:
        pre = cls._batch_decompress(state_dicts, [scheme] * len(state_dicts))
        return [cls.decompress(sd, scheme, _preweight=w, _prezp=z) for sd, (w, z) in zip(state_dicts, pre)]

    @classmethod
    def decompress_modules(cls, modules) -> None:
        from ...quantization.quant_args import QuantizationStatus
        from ...utils.module import get_direct_state_dict, replace_direct_state_dict

        modules = list(modules)
        sds = [get_direct_state_dict(m) for m in modules]
        pre = cls._batch_decompress(sds, [m.quantization_scheme for m in modules])
        for m, sd, (w, z) in zip(modules, sds, pre):
            replace_direct_state_dict(m, cls.decompress(sd, m.quantization_scheme, _preweight=w, _prezp=z))
            m.quantization_status = QuantizationStatus.DECOMPRESSED

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        """base.py:165-177"""
        ia = getattr(scheme, "input_activations", None)
        if ia is not None and enum_value(ia.type) == "float":
            return False
        w = getattr(scheme, "weights", None)
        return (
            module_type in COMPRESSIBLE_MODULE_TYPES
            and w is not None
            and 1 <= w.num_bits <= 8
            and enum_value(w.type) == "int"
        )
