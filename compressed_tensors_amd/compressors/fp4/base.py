"""FP4 pack-quantized codecs (reference compressors/nvfp4/base.py:27-139, mxfp4/base.py:27-65).

nvfp4-pack-quantized: E2M1 weights in groups of 16 under float8-e4m3 group scales and a float32 global scale;
mxfp4-pack-quantized: groups of 32 under E8M0 power-of-two scales.  The weight path of each direction is ONE fused
HIP kernel (`ct_fp4_quant_pack`: scale / global -> divide -> E2M1 rounding by v_cvt_scalef32_pk_fp4_f32 -> nibble
pack; `ct_fp4_unpack_dequant` reads the stored fp8 / E8M0 scale bytes directly); only the small scale tensors are
converted with torch ops."""
import torch

from ... import codec
from ...config import CompressionFormat
from ...quantization.quant_args import enum_value
from ...utils import getattr_chain
from ...quantization.quant_args import QuantizationStatus
from ...utils.module import direct_entry, swap_direct_entries
from ..base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor, symmetric_zp_keys

__all__ = ["NVFP4PackedCompressor", "MXFP4PackedCompressor"]


def _is_fp4(scheme, group_size) -> bool:
    w = getattr(scheme, "weights", None)
    return w is not None and int(w.num_bits) == 4 and enum_value(w.type) == "float" and getattr(w, "group_size", None) == group_size


@BaseCompressor.register(name=CompressionFormat.nvfp4_pack_quantized.value)
class NVFP4PackedCompressor(BaseCompressor):
    GROUP = 16

    @classmethod
    def compression_param_names(cls, scheme) -> tuple:
        """nvfp4/base.py:36-47"""
        names = ("weight_packed", "weight_scale", "weight_global_scale")
        if not getattr_chain(scheme, "weights.symmetric", True):
            names += ("weight_zero_point",)
        if not getattr_chain(scheme, "input_activations.dynamic", True):
            names += ("input_global_scale",)
        return names

    @classmethod
    def _compress_scale(cls, scale: torch.Tensor, weights) -> torch.Tensor:
        return scale.to(getattr(weights, "scale_dtype", None) or torch.float8_e4m3fn)

    @classmethod
    def _scale_kind(cls):
        return "f8e4m3"

    @classmethod
    def _decompress_scale(cls, scale: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        return scale.to(dtype)

    @classmethod
    def compress(cls, state_dict: dict, scheme) -> dict:
        """nvfp4/base.py:68-104"""
        state_dict = state_dict.copy()
        weight = state_dict.pop("weight")
        scale = state_dict.pop("weight_scale")
        global_scale = state_dict.get("weight_global_scale", None)
        if state_dict.get("weight_zero_point") is not None and not getattr_chain(scheme, "weights.symmetric", True):
            raise NotImplementedError("Asymmetric Quantization is not supported for FP4")
        # the packed weight and the stored scale from ONE launch where the layout allows (16-bit weights on the GPU, the scheme's usual scale dtype);
        # otherwise the scale conversion is the reference's expression
        sdt = getattr(scheme.weights, "scale_dtype", None) or (torch.float8_e4m3fn if cls.GROUP == 16 else torch.uint8)
        fused = codec.fp4_quantize_and_pack_stored(weight, scale, global_scale, group_size=cls.GROUP, scale_dtype=sdt)
        if fused is not None:
            state_dict["weight_packed"], state_dict["weight_scale"] = fused
        else:
            state_dict["weight_packed"] = codec.fp4_quantize_and_pack(weight, scale, global_scale, group_size=cls.GROUP)
            state_dict["weight_scale"] = cls._compress_scale(scale, scheme.weights)
        return cls._remove_symmetric_zp(state_dict, scheme)

    # ------------------------------------------------------------------ module loops (round 6)
    # `ModelCompressor` on an FP4 checkpoint was bound by the interpreter, not the GPU: 28 (NVFP4) / 44 us (MXFP4) of host work per module and
    # direction beside kernels of 2-30 us (a Llama-3-8B-shaped tree ran at 0.34 / 0.22 of the HBM peak).  The loops below do what
    # `compress_module` / `decompress_module` do for the usual module — the weight and its scale tensor from ONE launch, the parameter
    # dictionary rewritten as a delta (same resulting entries, in the same order, as replace_direct_state_dict leaves them) — and hand
    # everything else to the generic path.
    @classmethod
    def _native(cls, modules, direction: str):
        """the plain modules through the C++ loop (csrc/host/ct_hostpath.cpp: fp4_plan_compress / fp4_plan_decompress / fp4_finish) and ONE table launch per
        window (`ct_fp4_quant_pack_batch` / `ct_fp4_unpack_dequant_batch`) — the same results and the same dictionary delta as the loop below at ~3 instead of
        14 us of host work per module and without a launch per module; returns the modules it left for that loop"""
        from ... import _lib
        from ..pack_quantized.base import _launch_chunks

        modules = list(modules)
        hp = _lib.hostpath()
        if hp is None or not hasattr(hp, "fp4_plan_compress") or torch.nn.modules.module._global_parameter_registration_hooks or cls._native_group() is None:
            return modules
        want = torch.float8_e4m3fn if cls.GROUP == 16 else torch.uint8

        def info(scheme) -> int:
            wa = getattr(scheme, "weights", None)
            if wa is None or (getattr(wa, "scale_dtype", None) or want) is not want:
                return 0
            drop = 0
            for key in symmetric_zp_keys(scheme):
                drop |= {"weight_zero_point": 1, "input_zero_point": 2, "output_zero_point": 4}[key]
            return 1 | (drop << 1)

        compress = direction == "compress"
        status = QuantizationStatus.COMPRESSED if compress else QuantizationStatus.DECOMPRESSED
        codes = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}
        rest, pending = [], []
        for lo, hi in _launch_chunks(len(modules)):
            planned, back = hp.fp4_plan_compress(modules[lo:hi], info, cls.GROUP) if compress else hp.fp4_plan_decompress(modules[lo:hi], cls.GROUP)
            rest += back
            for (dev_index, code), (words, n, jobs, _zw, _zn) in planned.items():
                device = torch.device("cuda", dev_index) if dev_index >= 0 else torch.device("cpu")
                codec.launch_fp4_words(words, n, direction, device, cls.GROUP, codes[code & 15], codes[code >> 4])
                pending.append(jobs)
        for jobs in pending:
            hp.fp4_finish(jobs, status, compress)
        return rest

    @classmethod
    def _native_group(cls):
        """the group size when this class's module loops are the two FP4 formats' own (a subclass that overrides the codec keeps the Python loop)"""
        base = NVFP4PackedCompressor
        same = cls.compress.__func__ is base.compress.__func__ and cls.decompress.__func__ is base.decompress.__func__
        return cls.GROUP if same and cls.GROUP in (16, 32) else None

    @classmethod
    def compress_modules(cls, modules) -> None:
        modules = cls._native(modules, "compress")
        for m in modules:
            scheme = getattr(m, "quantization_scheme")
            w, sc, gs = direct_entry(m, "weight"), direct_entry(m, "weight_scale"), direct_entry(m, "weight_global_scale")
            fused = None
            if w is not None and sc is not None and (direct_entry(m, "weight_zero_point") is None or getattr_chain(scheme, "weights.symmetric", True)):
                sdt = getattr(scheme.weights, "scale_dtype", None) or (torch.float8_e4m3fn if cls.GROUP == 16 else torch.uint8)
                fused = codec.fp4_quantize_and_pack_stored(w.data, sc.data, None if gs is None else gs.data, group_size=cls.GROUP, scale_dtype=sdt)
            if fused is None:
                cls.compress_module(m)
                continue
            remove = ["weight", "weight_scale"] + [k for k in symmetric_zp_keys(scheme) if direct_entry(m, k) is not None]
            swap_direct_entries(m, remove, {"weight_packed": fused[0], "weight_scale": fused[1]}, status=QuantizationStatus.COMPRESSED)

    @classmethod
    def decompress_modules(cls, modules) -> None:
        modules = cls._native(modules, "decompress")
        for m in modules:
            packed, sc, gs = direct_entry(m, "weight_packed"), direct_entry(m, "weight_scale"), direct_entry(m, "weight_global_scale")
            if packed is None or sc is None or not packed.is_cuda or sc.device != packed.device:
                cls.decompress_module(m)
                continue
            weight, scale = codec.fp4_unpack_and_dequantize(packed.data, sc.data, None if gs is None else gs.data, group_size=cls.GROUP, scale_kind=cls._scale_kind(),
                                                            dtype=torch.bfloat16, return_scale=True)
            swap_direct_entries(m, ["weight_packed", "weight_scale"], {"weight_scale": scale, "weight": weight}, status=QuantizationStatus.DECOMPRESSED)

    @classmethod
    def decompress_many(cls, state_dicts, scheme) -> list:
        """`decompress` for several local-name state dicts of one scheme (the model-free converter: one safetensors shard, converters/ct_dequantizer.py:63-99):
        the tensors of the usual layout leave in ONE table launch (`ct_fp4_unpack_dequant_batch`: weights and bfloat16 scales), the rest one by one"""
        if cls._native_group() is None:
            return super().decompress_many(state_dicts, scheme)
        want = torch.float8_e4m3fn if cls.GROUP == 16 else torch.uint8
        out, words, keep, where = [None] * len(state_dicts), [], [], []
        device = None
        tail = (0,) * (codec._ITEM_WORDS - 11)
        for i, sd in enumerate(state_dicts):
            packed, sc, gs = sd.get("weight_packed"), sd.get("weight_scale"), sd.get("weight_global_scale")
            ok = (packed is not None and sc is not None and packed.is_cuda and packed.dtype is torch.uint8 and packed.dim() == 2 and packed.is_contiguous()
                  and packed.data_ptr() % 4 == 0 and sc.dtype is want and sc.device == packed.device and sc.is_contiguous() and (gs is not None) == (cls.GROUP == 16)
                  and (device is None or packed.device == device))
            if ok:
                rows, cols = int(packed.shape[0]), int(packed.shape[1]) * 2
                ok = rows > 0 and cols % cls.GROUP == 0 and (rows * cols) % 32 == 0 and tuple(sc.shape) == (rows, cols // cls.GROUP)
            if ok and gs is not None:
                ok = gs.dtype is torch.float32 and gs.numel() == 1 and gs.device == packed.device and gs.data_ptr() % 4 == 0
            if not ok:
                out[i] = cls.decompress(sd, scheme)
                continue
            device = packed.device
            weight = torch.empty((rows, cols), dtype=torch.bfloat16, device=device)  # unpack_fp4_from_uint8's default dtype (nvfp4/base.py:118-131)
            scale = torch.empty((rows, cols // cls.GROUP), dtype=torch.bfloat16, device=device)
            words += (packed.data_ptr(), sc.data_ptr(), 0 if gs is None else gs.data_ptr(), weight.data_ptr(), rows, cols, cls.GROUP, 0, 0, 0, scale.data_ptr(), *tail)
            keep.append((packed, sc, gs))
            new = dict(sd)
            del new["weight_packed"]
            new["weight_scale"] = scale
            new["weight"] = weight  # the entries in the order `decompress` leaves them
            out[i] = new
            where.append(i)
        if where:
            codec.launch_fp4_words(torch.tensor(words, dtype=torch.int64), len(where), "decompress", device, cls.GROUP)
        return out

    @classmethod
    def compress_rtn(cls, weight: torch.Tensor, scheme, global_scale=None) -> dict:
        """Round-to-nearest compression straight from the dense weight: generate_gparam (unless given), then observer +
        calculate_qparams + compress in ONE pass over the weight (codec.rtn_nvfp4_quantize_and_pack) — the state dict
        `compress` returns for {"weight", "weight_scale", "weight_global_scale"} of the min-max qparams."""
        if weight.dim() == 2 and weight.dtype in (torch.bfloat16, torch.float16) and weight.shape[1] % 32 == 0:
            packed, s8, gs = codec.rtn_nvfp4_quantize_and_pack(weight, global_scale)
            return {"weight_packed": packed, "weight_scale": s8.to(getattr(scheme.weights, "scale_dtype", None) or torch.float8_e4m3fn),
                    "weight_global_scale": gs}
        gs = codec.generate_gparam(weight) if global_scale is None else global_scale
        scale = codec.minmax_qparams_float(weight, kind="nvfp4", group_size=16, global_scale=gs)
        return cls.compress({"weight": weight, "weight_scale": scale, "weight_global_scale": gs}, scheme)

    @classmethod
    def decompress(cls, state_dict: dict, scheme) -> dict:
        """nvfp4/base.py:106-139: the weight comes back as bfloat16 (unpack_fp4_from_uint8's default), the scale as a
        bfloat16 tensor"""
        state_dict = state_dict.copy()
        packed = state_dict.pop("weight_packed")
        scale = state_dict.get("weight_scale")
        global_scale = state_dict.get("weight_global_scale", None)
        # the weight and the decompressed (bfloat16) scale from one launch
        state_dict["weight"], state_dict["weight_scale"] = codec.fp4_unpack_and_dequantize(
            packed, scale, global_scale, group_size=cls.GROUP, scale_kind=cls._scale_kind(), dtype=torch.bfloat16, return_scale=True)
        return state_dict

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        """nvfp4/base.py:130-139: FP4 with group_size 16"""
        return module_type in COMPRESSIBLE_MODULE_TYPES and _is_fp4(scheme, 16)


@BaseCompressor.register(name=CompressionFormat.mxfp4_pack_quantized.value)
class MXFP4PackedCompressor(NVFP4PackedCompressor):
    GROUP = 32

    @classmethod
    def compression_param_names(cls, scheme) -> tuple:
        """mxfp4/base.py:34-44: GROUP strategy, no global scale"""
        names = ("weight_packed", "weight_scale")
        if not getattr_chain(scheme, "weights.symmetric", True):
            names += ("weight_zero_point",)
        if not getattr_chain(scheme, "input_activations.dynamic", True):
            names += ("input_global_scale",)
        return names

    @classmethod
    def compress_rtn(cls, weight: torch.Tensor, scheme) -> dict:
        """Round-to-nearest compression straight from the dense weight (min-max qparams): the state dict `compress` returns
        for {"weight", "weight_scale"} with calculate_qparams of the weight's group min / max, in ONE pass over the weight
        (codec.rtn_mxfp4_quantize_and_pack) for 16-bit weights."""
        if weight.dim() == 2 and weight.dtype in (torch.bfloat16, torch.float16) and weight.shape[1] % 32 == 0:
            packed, code = codec.rtn_mxfp4_quantize_and_pack(weight)
            return {"weight_packed": packed, "weight_scale": code.to(getattr(scheme.weights, "scale_dtype", None) or torch.uint8)}
        scale = codec.minmax_qparams_float(weight, kind="mxfp4", group_size=32)
        return cls.compress({"weight": weight, "weight_scale": scale}, scheme)

    @classmethod
    def _compress_scale(cls, scale: torch.Tensor, weights) -> torch.Tensor:
        """mx_utils.py:18-31"""
        return codec.compress_mx_scale(scale, getattr(weights, "scale_dtype", None) or torch.uint8)

    @classmethod
    def _scale_kind(cls):
        return "e8m0"

    @classmethod
    def _decompress_scale(cls, scale: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """mx_utils.py:34-44"""
        return codec.decompress_mx_scale(scale).to(dtype)

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        """mxfp4/base.py:57-65: FP4 with group_size 32"""
        return module_type in COMPRESSIBLE_MODULE_TYPES and _is_fp4(scheme, 32)
