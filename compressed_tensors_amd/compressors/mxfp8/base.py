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

    # ------------------------------------------------------------------ module loops (round 6)
    # ModelCompressor's per-module loop in the C++ extension (csrc/host/ct_hostpath.cpp mx8_plan_compress / mx8_plan_decompress / mx8_finish): the weights of a
    # window of modules ride ONE launch of the 8-bit tables (float8 codes, groups of 32), their scale tensors ONE launch of ct_mx_scale_batch; whatever the C++
    # loop does not take goes through compress_module / decompress_module as before
    @classmethod
    def _native(cls, modules, direction: str):
        from ... import _lib
        from ...quantization.quant_args import QuantizationStatus
        from ..base import symmetric_zp_keys
        from ..pack_quantized.base import _launch_chunks

        modules = list(modules)
        hp = _lib.hostpath()
        base = MXFP8QuantizationCompressor
        if (hp is None or not hasattr(hp, "mx8_plan_compress") or torch.nn.modules.module._global_parameter_registration_hooks
                or cls.compress.__func__ is not base.compress.__func__ or cls.decompress.__func__ is not base.decompress.__func__):
            return modules

        def info(scheme) -> int:
            if not cls.can_compress(torch.nn.Linear, scheme):
                return 0
            drop = 0
            for key in symmetric_zp_keys(scheme):
                drop |= {"weight_zero_point": 1, "input_zero_point": 2, "output_zero_point": 4}[key]
            return 1 | (drop << 1)

        compress = direction == "compress"
        codes = {1: torch.float16, 2: torch.bfloat16}
        rest, pending = [], []
        for lo, hi in _launch_chunks(len(modules)):
            planned, back = hp.mx8_plan_compress(modules[lo:hi], info) if compress else hp.mx8_plan_decompress(modules[lo:hi])
            rest += back
            for (dev_index, code), (words, n, jobs, scale_words, scale_n) in planned.items():
                device = torch.device("cuda", dev_index) if dev_index >= 0 else torch.device("cpu")
                dtype = codes[code & 15]
                if compress:
                    codec.launch_q8_words(words, n, "compress", dtype, device, (code >> 4) & 15, 8)
                    codec.launch_mx_scale_words(scale_words, scale_n, "compress", device, dtype)
                else:  # the scales first: the weights' table reads their bfloat16 form
                    codec.launch_mx_scale_words(scale_words, scale_n, "decompress", device)
                    codec.launch_q8_words(words, n, "decompress", dtype, device, (code >> 4) & 15, 8)
                pending.append(jobs)
        status = QuantizationStatus.COMPRESSED if compress else QuantizationStatus.DECOMPRESSED
        for jobs in pending:
            hp.mx8_finish(jobs, status)
        return rest

    @classmethod
    def compress_modules(cls, modules) -> None:
        super().compress_modules(cls._native(modules, "compress"))

    @classmethod
    def decompress_modules(cls, modules) -> None:
        super().decompress_modules(cls._native(modules, "decompress"))

    @classmethod
    def decompress_many(cls, state_dicts, scheme) -> list:
        """`decompress` for several local-name state dicts (the model-free converter: one safetensors shard): the scale tensors of the usual layout through ONE
        launch of `ct_mx_scale_batch`, the weights through ONE of the 8-bit tables; the rest one by one"""
        base = MXFP8QuantizationCompressor
        if cls.compress.__func__ is not base.compress.__func__ or cls.decompress.__func__ is not base.decompress.__func__:
            return [cls.decompress(sd, scheme) for sd in state_dicts]
        out, words, swords, where = [None] * len(state_dicts), [], [], []
        device = None
        tail = (0,) * (codec._ITEM_WORDS - 7)
        for i, sd in enumerate(state_dicts):
            q, sc = sd.get("weight"), sd.get("weight_scale")
            ok = (q is not None and sc is not None and sd.get("weight_zero_point") is None and sd.get("weight_g_idx") is None and q.is_cuda and q.dim() == 2
                  and q.dtype is torch.float8_e4m3fn and q.is_contiguous() and q.data_ptr() % 16 == 0 and sc.dtype is torch.uint8 and sc.device == q.device
                  and sc.is_contiguous() and sc.dim() == 2 and (device is None or q.device == device))
            if ok:
                rows, cols = int(q.shape[0]), int(q.shape[1])
                ok = rows > 0 and cols % 32 == 0 and tuple(sc.shape) == (rows, cols // 32)
            if not ok:
                out[i] = cls.decompress(sd, scheme)
                continue
            device = q.device
            scale = torch.empty(sc.shape, dtype=torch.bfloat16, device=device)  # decompress_mx_scale: bfloat16, and so is the weight (mxfp8/base.py:74-101)
            weight = torch.empty((rows, cols), dtype=torch.bfloat16, device=device)
            words += (q.data_ptr(), scale.data_ptr(), 0, weight.data_ptr(), rows, cols, 32, *tail)
            swords += (sc.data_ptr(), 0, 0, scale.data_ptr(), sc.numel(), 1, 0, *tail)
            new = dict(sd, weight_scale=scale)
            del new["weight"]
            new["weight"] = weight  # the entries in the order `decompress` leaves them
            out[i] = new
            where.append(i)
        if where:
            codec.launch_mx_scale_words(torch.tensor(swords, dtype=torch.int64), len(where), "decompress", device)  # first: the weights' table reads the bfloat16 scales
            codec.launch_q8_words(torch.tensor(words, dtype=torch.int64), len(where), "decompress", torch.bfloat16, device, 1, 8)
        return out

    # the two hooks keep upstream's names: install() lets upstream's subclass call them
    @classmethod
    def _compress_scale(cls, scale: torch.Tensor, weights) -> torch.Tensor:
        return codec.compress_mx_scale(scale, getattr(weights, "scale_dtype", None) or torch.uint8)

    @classmethod
    def _decompress_scale(cls, scale: torch.Tensor) -> torch.Tensor:
        return codec.decompress_mx_scale(scale)
