"""pack-quantized codec (reference compressors/pack_quantized/base.py:35-177): INT weights of
1..8 bits packed densely into int32 words.

MI355X design: compress is ONE fused kernel (quantize -> clamp/round -> bitstream pack) and
decompress is ONE fused kernel (unpack -> dequantize); the int8 intermediate of the reference
never exists.  Algorithmic traffic per direction at W4A16 g128: 2 B/elem weight + 0.5 B/elem
packed + scales.
"""
import math

import torch

from ... import _lib, codec
from ...config import CompressionFormat
from ...quantization.quant_args import enum_value
from ...utils import getattr_chain
from ..base import COMPRESSIBLE_MODULE_TYPES, BaseCompressor, symmetric_zp_keys

__all__ = ["PackedQuantizationCompressor"]

PACK_ZP_STRATS = ("group", "channel")

def _hostpath():
    """the C++ host loop of the batched module paths (`_lib.hostpath`), or None when it has not been built: the Python loop below does
    the same work, four times slower per module"""
    return _lib.hostpath()


_DTYPE_OF_CODE = {1: torch.float16, 2: torch.bfloat16}


def _launch_chunks(n: int):
    """[lo, hi) slices of a module list for the batched launches: planning a 154-module table takes the host ~0.17 ms during which the
    GPU would idle, so a large list goes out as a short first table (the GPU starts after ~40 us) and two longer ones; a small list
    as one (every extra launch costs the host ~15 us and the device a ramp / tail of ~3 us)"""
    if n <= 64:
        return [(0, n)]
    a = 32
    b = a + (n - a) // 2
    return [(0, a), (a, b), (b, n)]


def _plain_w4_scheme(scheme):
    """(is an int4 group / channel scheme without activation arguments, group size or 0 for channel-wise, stores packed zero points): the
    schemes whose modules the C++ host loop takes (an activation scheme's extra zero-point names and everything else stay with the
    Python loop)"""
    wa = scheme.weights
    if (wa is None or getattr(scheme, "input_activations", None) is not None or getattr(scheme, "output_activations", None) is not None
            or int(wa.num_bits) != 4 or enum_value(getattr(wa, "type", "int")) != "int"):
        return False, -1, False
    st = enum_value(wa.strategy)
    asym = not wa.symmetric  # group / channel are exactly PACK_ZP_STRATS
    if st == "channel":
        return True, 0, asym
    if st == "group" and getattr(wa, "group_size", None):
        return True, int(wa.group_size), asym
    return False, -1, False


def _w8_info(scheme) -> int:
    """csrc/host/ct_hostpath.cpp w8_plan_compress / w8_plan_decompress: a symmetric weights-only int8 scheme (the W8A16 preset) -> group size | strategy << 25,
    else -1 (asymmetric 8-bit schemes store packed zero points and stay with the Python loop)"""
    wa = scheme.weights
    if (wa is None or getattr(scheme, "input_activations", None) is not None or getattr(scheme, "output_activations", None) is not None
            or int(wa.num_bits) != 8 or enum_value(getattr(wa, "type", "int")) != "int" or not wa.symmetric or enum_value(getattr(wa, "actorder", None)) == "group"):
        return -1
    st = enum_value(wa.strategy)
    if st not in ("tensor", "channel", "group"):
        return -1
    gs = int(getattr(wa, "group_size", None) or 0) if st == "group" else 0
    if not 0 <= gs < (1 << 20):
        return -1
    return gs | ({"tensor": 0, "channel": 1, "group": 2}[st] << 25)


def _native_w8(hp, modules, direction: str, status):
    """the 8-bit pack-quantized modules of `modules` through the C++ loop and the 8-bit tables' packed kind (one launch per window); returns the rest"""
    if not hasattr(hp, "w8_plan_compress") or not any(int(getattr(getattr(m.quantization_scheme, "weights", None), "num_bits", 0) or 0) == 8 for m in modules):
        return modules
    plan, finish = (hp.w8_plan_compress, hp.w4_finish_compress) if direction == "compress" else (hp.w8_plan_decompress, hp.w4_finish_decompress)
    rest, pending = [], []
    for lo, hi in _launch_chunks(len(modules)):
        planned, back = plan(modules[lo:hi], _w8_info)
        rest += back
        for (dev_index, code), (words, n, jobs, _zw, _zn) in planned.items():
            device = torch.device("cuda", dev_index) if dev_index >= 0 else torch.device("cpu")
            codec.launch_q8_words(words, n, direction, _DTYPE_OF_CODE[code & 15], device, (code >> 4) & 15, code >> 8)
            pending.append(jobs)
    for jobs in pending:
        finish(jobs, status)
    return rest


def _wb_info(scheme) -> int:
    """csrc/host/ct_hostpath.cpp wb_compress_modules / wb_decompress_modules: a symmetric weights-only int scheme, group / channel -> group size |
    strategy << 25 | num_bits << 28, else -1.  (The loop takes the modules no table does: the widths other than 4 and 8, and every width's modules with
    activation ordering — a `weight_g_idx` entry; it hands the others back.)"""
    wa = scheme.weights
    if (wa is None or getattr(scheme, "input_activations", None) is not None or getattr(scheme, "output_activations", None) is not None
            or enum_value(getattr(wa, "type", "int")) != "int"):
        return -1
    bits, st = int(wa.num_bits), enum_value(wa.strategy)
    if not 1 <= bits <= 8 or st not in ("channel", "group"):  # (= PACK_ZP_STRATS: an asymmetric scheme of these stores its zero points packed)
        return -1
    gs = int(getattr(wa, "group_size", None) or 0) if st == "group" else 0
    if not 0 <= gs < (1 << 20) or (st == "group" and gs <= 0):
        return -1
    return gs | ((0 if wa.symmetric else 1) << 24) | ({"channel": 1, "group": 2}[st] << 25) | (bits << 28)


def _native_wb(hp, modules, direction: str, status):
    """the modules of the other word widths on the current GPU through the C++ loop (one launch per module, by address); returns the rest"""
    if (not hasattr(hp, "wb_compress_modules") or not torch.cuda.is_available()
            or not any(int(getattr(getattr(m.quantization_scheme, "weights", None), "num_bits", 4) or 4) != 4 or m._parameters.get("weight_g_idx") is not None
                       for m in modules)):
        return modules
    fn = hp.wb_compress_modules if direction == "compress" else hp.wb_decompress_modules
    # per GPU (an HF `device_map` model spreads its modules over several in one process): the loop launches on the device it is given, which is made
    # current for the call; a module whose tensors are elsewhere comes back in the rest
    name = "weight" if direction == "compress" else "weight_packed"
    by_device, rest = {}, []
    for m in modules:
        t = m._parameters.get(name)
        if t is not None and t.is_cuda:
            by_device.setdefault(t.device.index if t.device.type == "cuda" else None, []).append(m)  # (None: the CPU stand-ins of tests/test_host_logic.py)
        else:
            rest.append(m)
    for index, ms in by_device.items():
        if index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
            rest += fn(ms, _wb_info, dev.index, int(_lib.stream_on(dev)), status)
            continue
        dev = torch.device("cuda", index)
        with torch.cuda.device(dev):
            rest += fn(ms, _wb_info, index, int(_lib.stream_on(dev)), status)
    return rest


_ASYMMETRIC = 1 << 40  # csrc/host/ct_hostpath.cpp: kAsymmetric


def _compress_info(scheme) -> int:
    ok, group, asym = _plain_w4_scheme(scheme)
    return (group + (_ASYMMETRIC if asym else 0)) if ok else -1


def _decompress_info(scheme) -> int:
    ok, _, asym = _plain_w4_scheme(scheme)
    return (2 if asym else 1) if ok else 0


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
        stores_zp = not weights.symmetric and enum_value(weights.strategy) in PACK_ZP_STRATS
        if _prepacked is not None:
            state_dict["weight_packed"] = _prepacked  # produced by the batched launch of compress_modules
        else:
            fused = None
            if stores_zp and g_idx is None and _prezp is None and zero_point is not None:
                # the weight words AND the stored form of the zero points from ONE launch (ct_quant_pack_w4_zp; None: not its layout)
                fused = codec.quantize_and_pack_with_zp(weight, scale, zero_point, num_bits=int(weights.num_bits), strategy=enum_value(weights.strategy),
                                                        group_size=getattr(weights, "group_size", None))
            if fused is not None:
                state_dict["weight_packed"], _prezp = fused
            else:
                state_dict["weight_packed"] = codec.quantize_and_pack(weight, scale, zero_point, g_idx=g_idx, **_layout_kwargs(weights))
        state_dict["weight_shape"] = torch.tensor(weight.shape)  # int64, CPU: as upstream (:105)

        if stores_zp:
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
            if _prezp is None and _preweight is None and g_idx is None and packed.is_cuda:
                # ONE launch: the kernel reads the zero points in their stored form and writes the unpacked int8 form back beside the
                # weight (ct_unpack_dequant_w4_zp; None: not its layout — groups of 128, cols % 512 == 0)
                fused = codec.unpack_and_dequantize_with_zp(packed, shape, scale, zero_point, num_bits=int(weights.num_bits))
                if fused is not None:
                    _preweight, _prezp = fused
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
        """One launch for every eligible module (int4, 16-bit, group / channel scales), the rest one by one.  The modules end in
        exactly the state `compress_module` leaves them in.

        Host side (bench.py `tinyllama_checkpoint.api`): the table is built from the modules' own entries (no state-dict copies), the
        launch is issued BEFORE any module is touched, and the per-module bookkeeping — drop `weight` (and the zero point of a
        symmetric scheme), add `weight_packed` / `weight_shape` — then runs under the kernel as a delta (`swap_direct_entries`)."""
        from ...quantization.quant_args import QuantizationStatus
        from ...utils.module import swap_direct_entries

        hp = _hostpath()
        if hp is not None and not torch.nn.modules.module._global_parameter_registration_hooks:
            # the plain case — int4 group / channel, parameters only, nn.Module's own attribute hooks — in C++: table rows, output allocations
            # and, after the launch, the parameter dictionaries; whatever it does not take comes back in `modules`
            modules = _native_w8(hp, list(modules), "compress", QuantizationStatus.COMPRESSED)
            modules = _native_wb(hp, modules, "compress", QuantizationStatus.COMPRESSED)
            rest, pending = [], []
            for lo, hi in _launch_chunks(len(modules)):  # the first launch leaves after a fifth of the planning, not after all of it
                planned, back = hp.w4_plan_compress(modules[lo:hi], _compress_info)  # (asked once per distinct scheme object and chunk)
                rest += back
                for (dev_index, code), (words, n, jobs, zp_words, zp_n) in planned.items():
                    device = torch.device("cuda", dev_index) if dev_index >= 0 else torch.device("cpu")
                    codec.launch_w4_words(words, n, "compress", _DTYPE_OF_CODE[code], device)
                    # (the asymmetric modules' zero points — pack_to_int32(zp, 4, packed_dim=0) — are written by tail workgroups of that launch;
                    # zp_n is 0 since round 6 and the call below a no-op, kept for a host extension built from an older source)
                    codec.launch_zp4_words(zp_words, zp_n, "pack", device)
                    pending.append(jobs)
            for jobs in pending:  # the parameter dictionaries, under the kernels
                hp.w4_finish_compress(jobs, QuantizationStatus.COMPRESSED)
            modules = rest

        batches = {}  # (device, dtype) -> (entries, jobs): one table and one launch per GPU and weight dtype
        rest = []
        schemes = {}  # id(scheme) -> (group or 0 for channel, asymmetric with packed zero points) | None: what only depends on the scheme
        f16 = (torch.bfloat16, torch.float16)
        for m in modules:
            scheme = m.quantization_scheme
            info = schemes.get(id(scheme), 0)
            if info == 0:
                wa = scheme.weights
                st = enum_value(wa.strategy)
                info = None
                if int(wa.num_bits) == 4 and enum_value(getattr(wa, "type", "int")) == "int" and st in ("group", "channel"):
                    g = 0 if st == "channel" else int(getattr(wa, "group_size", 0) or -1)
                    if g >= 0 and g % 32 == 0:
                        info = (g, not wa.symmetric)
                schemes[id(scheme)] = info
            params, buffers = m._parameters, m._buffers
            w = params.get("weight")
            if w is None:
                w = buffers.get("weight")
            scale = params.get("weight_scale")
            if scale is None:
                scale = buffers.get("weight_scale")
            zp = params.get("weight_zero_point")
            if zp is None:
                zp = buffers.get("weight_zero_point")
            # the conditions of codec.w4_batch_eligible, on the module's own entries
            ok = (info is not None and w is not None and scale is not None and w.dim() == 2 and w.dtype in f16 and scale.dtype is w.dtype
                  and w.is_cuda and w.is_contiguous() and w.data_ptr() % 16 == 0
                  and scale.is_contiguous() and scale.data_ptr() % 16 == 0 and scale.device == w.device
                  and "weight_g_idx" not in params and "weight_g_idx" not in buffers)
            if ok:
                rows, cols = w.shape
                group = info[0] or cols
                ok = rows > 0 and cols % 32 == 0 and cols % group == 0 and scale.shape == (rows, cols // group)
                if ok and zp is not None:
                    ok = (zp.dtype is torch.int8 and zp.shape == scale.shape and zp.is_contiguous() and zp.data_ptr() % 16 == 0
                          and zp.device == w.device)
            if not ok:
                rest.append(m)
                continue
            packed = torch.empty((rows, cols // 8), dtype=torch.int32, device=w.device)
            entries, jobs = batches.setdefault((w.device, w.dtype), ([], []))
            zpp = None
            if info[1] and enum_value(scheme.weights.strategy) in PACK_ZP_STRATS:
                assert zp is not None, "Asymmetric quant requires zero-point values"
                # the stored form of the zero points (pack_to_int32(zp, 4, packed_dim=0)): written by tail workgroups of the SAME launch
                zpp = torch.empty((math.ceil(rows * 4 / 32), zp.shape[1]), dtype=torch.int32, device=w.device)
            entries.append((w, scale, zp, packed, rows, cols, group, zpp))
            jobs.append((m, scheme, packed, zp, (rows, cols)))
        for (device, dtype), (entries, jobs) in batches.items():
            codec.W4Batch(entries, "compress", dtype).launch()
            zps = {i: (e[2], e[7]) for i, e in enumerate(entries) if e[7] is not None}
            # from here on the host works under the kernels
            shape_of = {}
            for i, (m, scheme, packed, zp, shape) in enumerate(jobs):
                base = shape_of.get(shape)
                if base is None:
                    base = shape_of[shape] = torch.tensor(shape)  # int64, CPU: as upstream (:105)
                add = {"weight_packed": packed, "weight_shape": base.clone()}
                remove = ["weight"]
                wa = scheme.weights
                if not wa.symmetric and enum_value(wa.strategy) in PACK_ZP_STRATS:
                    add["weight_zero_point"] = zps[i][1] if i in zps else codec.pack_to_int32(zp.data.to(torch.int8), wa.num_bits, packed_dim=0)
                remove += symmetric_zp_keys(scheme)
                swap_direct_entries(m, remove, add, QuantizationStatus.COMPRESSED)
        for m in rest:
            cls._delta_module(m, "compress")

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
            shape = tuple(shape_t.tolist())
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
            if asym and codec.w4_packed_zp_readable(shape[1], group) and shape[0] * (shape[1] // 8) < 1 << 31 and zp_packed.data_ptr() % 16 == 0:
                # the weights' launch reads the stored zero points itself and writes `zp` (the unpacked int8 form) from its tail workgroups
                entries.append((packed, scale, zp, out, shape[0], shape[1], group, zp_packed))
            else:
                entries.append((packed, scale, zp, out, shape[0], shape[1], group))
                if asym:
                    zps.append((zp_packed, zp))
            slots.append((i, out, zp))
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
        """one launch (+ one for the packed zero points) for every eligible module; the modules end in exactly the state
        `decompress_module` leaves them in: `weight_packed` replaced by `weight`, an asymmetric scheme's zero point unpacked to
        int8, `weight_shape` kept (base.py:116-163)"""
        from ...quantization.quant_args import QuantizationStatus
        from ...utils.module import direct_entry, swap_direct_entries

        modules = list(modules)
        hp = _hostpath()
        if hp is not None and not torch.nn.modules.module._global_parameter_registration_hooks:
            modules = _native_w8(hp, modules, "decompress", QuantizationStatus.DECOMPRESSED)
            modules = _native_wb(hp, modules, "decompress", QuantizationStatus.DECOMPRESSED)
            rest, pending = [], []
            for lo, hi in _launch_chunks(len(modules)):
                planned, back = hp.w4_plan_decompress(modules[lo:hi], _decompress_info)
                rest += back
                for (dev_index, code), (words, n, jobs, zp_words, zp_n) in planned.items():
                    device = torch.device("cuda", dev_index) if dev_index >= 0 else torch.device("cpu")
                    codec.launch_zp4_words(zp_words, zp_n, "unpack", device)  # first: the weights' table points at the unpacked zero points
                    codec.launch_w4_words(words, n, "decompress", _DTYPE_OF_CODE[code], device)
                    pending.append(jobs)
            for jobs in pending:
                hp.w4_finish_decompress(jobs, QuantizationStatus.DECOMPRESSED)
            modules = rest
        names = ("weight_packed", "weight_scale", "weight_shape", "weight_zero_point", "weight_g_idx")
        sds = [{k: t for k in names if (t := direct_entry(m, k)) is not None} for m in modules]
        pre = PackedQuantizationCompressor._batch_decompress(sds, [m.quantization_scheme for m in modules])  # (`cls` may be install()'s subclass of the UPSTREAM codec)
        for m, (w, z) in zip(modules, pre):
            if w is None:
                cls._delta_module(m, "decompress")
                continue
            add = {"weight": w}
            if z is not None:
                add["weight_zero_point"] = z
            swap_direct_entries(m, ("weight_packed",), add, QuantizationStatus.DECOMPRESSED)

    _DELTA_NAMES = ("weight", "weight_packed", "weight_scale", "weight_shape", "weight_zero_point", "weight_g_idx")

    @classmethod
    def _delta_module(cls, m, direction: str) -> None:
        """`compress_module` / `decompress_module` for a module no table takes (2 / 3 / 6-bit words, activation ordering, asymmetric 8-bit, odd layouts): the
        codec itself on the module's own weight entries, the result written back as a delta (`swap_direct_entries`: same names, order and kinds as
        replace_direct_state_dict leaves, compressors/base.py:95-131) — 31-35 us of host work per module through the generic path, ~20 this way.  Anything
        but parameters on a GPU goes the generic way."""
        from ...quantization.quant_args import QuantizationStatus
        from ...utils.module import swap_direct_entries
        from ..base import symmetric_zp_keys

        params = m._parameters
        sd = {k: params[k].data for k in cls._DELTA_NAMES if params.get(k) is not None}  # (.data: what get_direct_state_dict hands the codec)
        src = sd.get("weight" if direction == "compress" else "weight_packed")
        if src is None or not src.is_cuda or any(k in m._buffers for k in cls._DELTA_NAMES):
            return cls.compress_module(m) if direction == "compress" else cls.decompress_module(m)
        scheme = m.quantization_scheme
        new = cls.compress(sd, scheme) if direction == "compress" else cls.decompress(sd, scheme)
        remove = [k for k in sd if k not in new]
        add = {k: v for k, v in new.items() if v is not sd.get(k)}
        if direction == "compress":
            remove += [k for k in symmetric_zp_keys(scheme) if k not in remove]
            status = QuantizationStatus.COMPRESSED
        else:
            status = QuantizationStatus.DECOMPRESSED
        swap_direct_entries(m, remove, add, status)

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
