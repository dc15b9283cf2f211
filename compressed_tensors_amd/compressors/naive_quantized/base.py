"""naive-quantized / int-quantized / float-quantized codecs
(reference compressors/naive_quantized/base.py:27-164).  INT weights are quantized to int8 by
the HIP quantize kernel and dequantized by the HIP dequantize kernel; FLOAT 8-bit weights become
float8_e4m3fn through the same kernels (clamp to +-448, v_cvt_pk_fp8_f32).

A list of modules / state dicts (`compress_modules`, `decompress_modules`, `decompress_many`: what ModelCompressor and the
model-free converter call) goes through ONE launch per direction, device and dtype (`ct_q8_quant_batch` /
`ct_q8_dequant_batch`) for every eligible tensor (16-bit weight and scale of one dtype, tensor / channel / group scales, int8 or no
zero point); the rest, one by one.  The state dicts are identical to the per-module path's."""
import torch

from ... import codec
from ...config import CompressionFormat
from ...quantization.quant_args import enum_value
from ...utils import getattr_chain
from ..base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor

__all__ = ["NaiveQuantizationCompressor", "IntQuantizationCompressor", "FloatQuantizationCompressor"]

_DTYPE_OF_CODE = {1: torch.float16, 2: torch.bfloat16}
_STRATEGY_CODE = {"tensor": 0, "channel": 1, "group": 2, "block": 3}


def _q8_compress_info(scheme) -> int:
    """what the C++ host loop needs to know of a scheme (csrc/host/ct_hostpath.cpp, q8_plan_compress): group size, num_bits, FLOAT, strategy and the
    zero points a symmetric scheme does not store — or -1 for a scheme whose modules stay with the Python loop"""
    from ..base import symmetric_zp_keys

    wa = getattr(scheme, "weights", None)
    if wa is None:
        return -1
    qtype, st = enum_value(getattr(wa, "type", "int")), enum_value(wa.strategy)
    bits = int(wa.num_bits)
    if st not in _STRATEGY_CODE or qtype not in ("int", "float") or not 1 <= bits <= 8 or (qtype == "float" and bits != 8):
        return -1
    if enum_value(getattr(wa, "actorder", None)) == "group":
        return -1
    gs = int(getattr(wa, "group_size", None) or 0) if st == "group" else 0
    bh = 0
    if st == "block":  # block width in the group-size field, block height behind the drop mask
        bs = getattr(wa, "block_structure", None)
        if bs is None or len(bs) != 2:
            return -1
        bh, gs = int(bs[0]), int(bs[1])
    if not 0 <= gs < (1 << 20) or not 0 <= bh < (1 << 20):
        return -1
    drop = 0
    for key in symmetric_zp_keys(scheme):
        drop |= {"weight_zero_point": 1, "input_zero_point": 2, "output_zero_point": 4}[key]
    return gs | (bits << 20) | ((qtype == "float") << 24) | (_STRATEGY_CODE[st] << 25) | (drop << 27) | (bh << 30)


def _q8_decompress_info(scheme) -> int:
    return 1 if getattr(scheme, "weights", None) is not None else 0


def _native_q8(modules, direction: str, status):
    """the plain modules of `modules` through the C++ host loop (table rows, output allocations, launches in windows, the parameter dictionaries under the
    kernels); returns the modules it did not take.  None of it when the extension is not built or a global parameter-registration hook is installed."""
    from ... import _lib
    from ..pack_quantized.base import _launch_chunks

    hp = _lib.hostpath()
    if hp is None or not hasattr(hp, "q8_plan_compress") or torch.nn.modules.module._global_parameter_registration_hooks:
        return modules
    plan, info = (hp.q8_plan_compress, _q8_compress_info) if direction == "compress" else (hp.q8_plan_decompress, _q8_decompress_info)
    rest, pending = [], []
    for lo, hi in _launch_chunks(len(modules)):
        planned, back = plan(modules[lo:hi], info)
        rest += back
        for (dev_index, code), (words, n, jobs, _zw, _zn) in planned.items():
            device = torch.device("cuda", dev_index) if dev_index >= 0 else torch.device("cpu")
            codec.launch_q8_words(words, n, direction, _DTYPE_OF_CODE[code & 15], device, (code >> 4) & 15, code >> 8)
            pending.append(jobs)
    for jobs in pending:
        hp.q8_finish(jobs, status)
    return rest


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

    # ------------------------------------------------------------------ batched module paths
    @classmethod
    def _batch_compress(cls, state_dicts, schemes):
        """quantized weights of every eligible state dict from one launch per (device, dtype, kind, bits); None for the others"""
        outs, batches = [None] * len(state_dicts), {}
        for i, (sd, scheme) in enumerate(zip(state_dicts, schemes)):
            w, scale, zp = sd.get("weight"), sd.get("weight_scale"), sd.get("weight_zero_point")
            wa = scheme.weights
            qtype, st = enum_value(getattr(wa, "type", "int")), enum_value(wa.strategy)
            if w is None or not w.is_cuda or not w.is_contiguous() or w.data_ptr() % 16 or st not in ("tensor", "channel", "group", "block"):
                continue
            if qtype == "float" and int(wa.num_bits) != 8:
                continue
            # a FLOAT scheme's zero point (float8, all zeros after calibration) is present in the usual flow and adds (-0.0 -> +0.0): its own batch kind,
            # whose kernels read the zero points as the float8 values they are (round 6; these modules took one launch each before)
            f8z = qtype == "float" and zp is not None and zp.dtype == torch.float8_e4m3fn
            if qtype == "float" and zp is not None and not f8z:
                continue
            group = codec.q8_batch_group(w.shape, w.dtype, scale, zp, device=w.device, strategy=st, group_size=getattr(wa, "group_size", None),
                                         g_idx=sd.get("weight_g_idx"), f8_zero_point=f8z, block_structure=getattr(wa, "block_structure", None))
            if group is None:
                continue
            out = torch.empty(w.shape, dtype=wa.pytorch_dtype(), device=w.device)
            if out.element_size() != 1:
                continue
            key = (w.device, w.dtype, ("fp8z" if f8z else "fp8") if qtype == "float" else "int8", int(wa.num_bits))
            entries = batches.setdefault(key, [])
            entries.append((w, scale, zp, out, w.shape[0], w.shape[1], group))
            outs[i] = out
            if len(entries) >= cls._BATCH_CHUNK:  # the GPU starts on this window while the interpreter plans the next one
                codec.W4Batch(entries, "compress", key[1], kind=key[2], bits=key[3]).launch()
                entries.clear()
        for (_, dtype, kind, bits), entries in batches.items():
            if entries:
                codec.W4Batch(entries, "compress", dtype, kind=kind, bits=bits).launch()
        return outs

    # modules per table launch of the interpreter-planned 8-bit batches (round 6): planning a module costs ~3.5 us of host time BEFORE its launch can go out;
    # with one table for a whole 8B-shaped checkpoint the GPU idled for 0.4 ms per direction (0.63 of the HBM peak through ModelCompressor — no better than one
    # launch per module), with windows the first launch leaves after ~0.1 ms and the rest is planned under the kernels
    _BATCH_CHUNK = 32

    @classmethod
    def _batch_decompress(cls, state_dicts):
        outs, batches = [None] * len(state_dicts), {}
        for i, sd in enumerate(state_dicts):
            q, scale, zp = sd.get("weight"), sd.get("weight_scale"), sd.get("weight_zero_point")
            if q is None or scale is None or not q.is_cuda or not q.is_contiguous() or q.data_ptr() % 16 or q.dim() != 2:
                continue
            kind = "int8" if q.dtype == torch.int8 else "fp8" if q.dtype == torch.float8_e4m3fn else None
            f8z = kind == "fp8" and zp is not None and zp.dtype == torch.float8_e4m3fn
            if kind is None or (kind == "fp8" and zp is not None and not f8z):
                continue
            if f8z:
                kind = "fp8z"
            group = codec.q8_batch_group(q.shape, scale.dtype, scale, zp, device=q.device, g_idx=sd.get("weight_g_idx"), f8_zero_point=f8z)
            if group is None:
                continue
            out = torch.empty(q.shape, dtype=scale.dtype, device=q.device)
            key = (q.device, scale.dtype, kind)
            entries = batches.setdefault(key, [])
            entries.append((q, scale, zp, out, q.shape[0], q.shape[1], group))
            outs[i] = out
            if len(entries) >= cls._BATCH_CHUNK:
                codec.W4Batch(entries, "decompress", key[1], kind=kind).launch()
                entries.clear()
        for (_, dtype, kind), entries in batches.items():
            if entries:
                codec.W4Batch(entries, "decompress", dtype, kind=kind).launch()
        return outs

    @classmethod
    def _owns_codec(cls) -> bool:
        """a subclass that overrides compress / decompress (mxfp8: scale conversion around them) keeps the per-module loop"""
        base = NaiveQuantizationCompressor
        return cls.compress.__func__ is base.compress.__func__ and cls.decompress.__func__ is base.decompress.__func__

    @classmethod
    def compress_modules(cls, modules) -> None:
        from ...quantization.quant_args import QuantizationStatus
        from ...utils.module import direct_entry, swap_direct_entries
        from ..base import symmetric_zp_keys

        modules = list(modules)
        if not cls._owns_codec():
            return super().compress_modules(modules)
        modules = _native_q8(modules, "compress", QuantizationStatus.COMPRESSED)
        names = ("weight", "weight_scale", "weight_zero_point", "weight_g_idx")
        sds = [{k: t for k in names if (t := direct_entry(m, k)) is not None} for m in modules]  # the modules' own entries: no state-dict copies
        pre = cls._batch_compress(sds, [m.quantization_scheme for m in modules])
        for m, q in zip(modules, pre):
            if q is None:
                cls.compress_module(m)
                continue
            # what replace_direct_state_dict(module, compress(state_dict)) leaves: the codes in `weight` — which `compress` pops and re-adds, so it ends
            # up BEHIND the entries that stay (naive_quantized/base.py:62-100) —, no zero point for a symmetric scheme
            swap_direct_entries(m, [*symmetric_zp_keys(m.quantization_scheme), "weight"], {"weight": q}, QuantizationStatus.COMPRESSED)

    @classmethod
    def decompress_many(cls, state_dicts, scheme) -> list:
        if not cls._owns_codec():
            return super().decompress_many(state_dicts, scheme)
        pre = cls._batch_decompress(state_dicts)
        out = []
        for sd, w in zip(state_dicts, pre):
            if w is None:
                out.append(cls.decompress(sd, scheme))
            else:
                new = dict(sd)
                new["weight"] = w
                out.append(new)
        return out

    @classmethod
    def decompress_modules(cls, modules) -> None:
        from ...quantization.quant_args import QuantizationStatus
        from ...utils.module import direct_entry, swap_direct_entries

        modules = list(modules)
        if not cls._owns_codec():
            return super().decompress_modules(modules)
        modules = _native_q8(modules, "decompress", QuantizationStatus.DECOMPRESSED)
        names = ("weight", "weight_scale", "weight_zero_point", "weight_g_idx")
        sds = [{k: t for k in names if (t := direct_entry(m, k)) is not None} for m in modules]
        pre = cls._batch_decompress(sds)
        for m, w in zip(modules, pre):
            if w is None:
                cls.decompress_module(m)
                continue
            swap_direct_entries(m, ("weight",), {"weight": w}, QuantizationStatus.DECOMPRESSED)  # (popped and re-added by `decompress`: last, as upstream leaves it)

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
