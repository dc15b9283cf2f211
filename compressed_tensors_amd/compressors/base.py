"""Compressor plug-in surface, mirrored from the reference (compressors/base.py:34-219).

`BaseCompressor` subclasses are never instantiated: they are looked up by their wire-format
string (`BaseCompressor.get_value_from_registry("pack-quantized")`) and used through
classmethods on *local-name* state dicts (`weight`, `weight_scale`, ...).  Inputs are never
mutated; untouched tensors are returned by identity.
"""
from abc import ABC
from typing import Optional

import torch

from ..config import CompressionFormat
from ..quantization.quant_args import QuantizationStatus, is_scheme
from ..registry import RegistryMixin
from ..utils.module import get_direct_state_dict, replace_direct_state_dict

__all__ = ["BaseCompressor", "symmetric_zp_keys", "compress_module", "decompress_module", "compress_modules", "decompress_modules", "COMPRESSIBLE_MODULE_TYPES"]

# reference compressors/base.py:31
COMPRESSIBLE_MODULE_TYPES = (torch.nn.Linear, torch.nn.Embedding)


_ZP_OF_ARGS = (("input_activations", "input_zero_point"), ("weights", "weight_zero_point"), ("output_activations", "output_zero_point"))


def symmetric_zp_keys(scheme) -> list:
    """the zero-point names that a symmetric scheme does not store (compressors/base.py:147-167)"""
    keys = []
    for args_name, key in _ZP_OF_ARGS:
        args = getattr(scheme, args_name, None)
        if args is not None and getattr(args, "symmetric", False):
            keys.append(key)
    return keys


class BaseCompressor(RegistryMixin, ABC):
    @classmethod
    def compression_param_names(cls, scheme) -> tuple:
        raise NotImplementedError(
            f"{cls.__name__} does not implement the classmethod compression_param_names interface"
        )

    @classmethod
    def compress(cls, state_dict: dict, scheme) -> dict:
        raise NotImplementedError(f"{cls.__name__} does not implement the classmethod compress interface")

    @classmethod
    def decompress(cls, state_dict: dict, scheme) -> dict:
        raise NotImplementedError(f"{cls.__name__} does not implement the classmethod decompress interface")

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        raise NotImplementedError(f"{cls.__name__} does not implement match")

    @classmethod
    def compress_module(cls, module: torch.nn.Module) -> None:
        """compressors/base.py:95-112"""
        scheme = getattr(module, "quantization_scheme")
        state_dict = get_direct_state_dict(module)
        replace_direct_state_dict(module, cls.compress(state_dict, scheme))
        module.quantization_status = QuantizationStatus.COMPRESSED

    @classmethod
    def decompress_module(cls, module: torch.nn.Module) -> None:
        """compressors/base.py:114-131"""
        scheme = getattr(module, "quantization_scheme")
        state_dict = get_direct_state_dict(module)
        replace_direct_state_dict(module, cls.decompress(state_dict, scheme))
        module.quantization_status = QuantizationStatus.DECOMPRESSED

    @classmethod
    def compress_modules(cls, modules) -> None:
        """compress several modules of this format; codecs may override to batch their launches
        (the reference loops, model_compressor.py:167-169)"""
        for m in modules:
            cls.compress_module(m)

    @classmethod
    def decompress_modules(cls, modules) -> None:
        for m in modules:
            cls.decompress_module(m)

    @classmethod
    def decompress_many(cls, state_dicts, scheme) -> list:
        """decompress several local-name state dicts of one scheme (the model-free path: one safetensors
        shard, converters/ct_dequantizer.py:63-99); codecs may override to batch their launches"""
        return [cls.decompress(sd, scheme) for sd in state_dicts]

    @classmethod
    def _remove_symmetric_zp(cls, state_dict: dict, scheme) -> dict:
        """compressors/base.py:147-167: vLLM cannot load zero points of symmetric schemes"""
        for key in symmetric_zp_keys(scheme):
            state_dict.pop(key, None)
        return state_dict


def _resolve_format(module, scheme, format):
    from .format import infer_module_format

    fmt = format or getattr(scheme, "format", None) or infer_module_format(type(module), scheme)
    fmt = CompressionFormat(getattr(fmt, "value", fmt))
    try:
        scheme.format = fmt
    except Exception:  # pydantic schemes validate assignment; the string value is always accepted
        scheme.format = fmt.value
    return fmt


def compress_module(module: torch.nn.Module, format: Optional[CompressionFormat] = None):
    """compressors/base.py:170-193"""
    scheme = getattr(module, "quantization_scheme", None)
    if not is_scheme(scheme):
        return
    fmt = _resolve_format(module, scheme, format)
    BaseCompressor.get_value_from_registry(fmt.value).compress_module(module)


def _by_format(modules, format):
    """modules grouped by the wire format their scheme resolves to; the resolution (and the `scheme.format` write-back of
    compress_module, compressors/base.py:186-190) happens once per (scheme object, module type), not once per module"""
    groups, seen = {}, {}
    for m in modules:
        scheme = getattr(m, "quantization_scheme", None)
        key = (id(scheme), type(m))
        fmt = seen.get(key)
        if fmt is None:
            if not is_scheme(scheme):
                continue
            fmt = seen[key] = _resolve_format(m, scheme, format).value
        group = groups.get(fmt)
        if group is None:
            group = groups[fmt] = []
        group.append(m)
    return groups


def compress_modules(modules, format: Optional[CompressionFormat] = None):
    """compress_module over a list, grouped by format so that a codec can batch its kernel launches"""
    for fmt, ms in _by_format(modules, format).items():
        BaseCompressor.get_value_from_registry(fmt).compress_modules(ms)


def decompress_modules(modules, format: Optional[CompressionFormat] = None):
    for fmt, ms in _by_format(modules, format).items():
        BaseCompressor.get_value_from_registry(fmt).decompress_modules(ms)


def decompress_module(module: torch.nn.Module, format: Optional[CompressionFormat] = None):
    """compressors/base.py:196-219"""
    scheme = getattr(module, "quantization_scheme", None)
    if not is_scheme(scheme):
        return
    fmt = _resolve_format(module, scheme, format)
    BaseCompressor.get_value_from_registry(fmt.value).decompress_module(module)
