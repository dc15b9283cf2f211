"""quantize / dequantize / fake_quantize with the reference's signatures
(quantization/lifecycle/forward.py:36-181), executed by the HIP kernels.

Scope: INT quantization (num_bits 1..8), strategies tensor / channel / token / group / block,
optional activation ordering (g_idx).  FLOAT types (fp8 / fp4) and `global_scale` belong to
the FP4/MX formats, which SURVEY.md §8 marks out of scope; they raise NotImplementedError.
"""
from typing import Optional

import torch

from .. import codec
from .quant_args import enum_value

__all__ = ["quantize", "dequantize", "fake_quantize", "calculate_range"]


def _int_args(args, global_scale):
    if global_scale is not None:
        raise NotImplementedError("global_scale (FP4 tensor-group quantization) is not on the MI355X hot path")
    if enum_value(getattr(args, "type", "int")) != "int":
        raise NotImplementedError("only INT quantization is implemented by the MI355X hot path")
    return dict(
        num_bits=int(args.num_bits),
        strategy=enum_value(args.strategy),
        group_size=getattr(args, "group_size", None),
        block_structure=getattr(args, "block_structure", None),
    )


def calculate_range(quantization_args, device=None):
    """quantization/utils/helpers.py:198-226 for INT types, as python floats (the kernels take
    immediates; no device tensors are created)."""
    if enum_value(getattr(quantization_args, "type", "int")) != "int":
        raise NotImplementedError("only INT ranges are implemented")
    bit_range = 2.0 ** quantization_args.num_bits
    return -bit_range / 2, bit_range / 2 - 1


@torch.no_grad()
def quantize(x: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor], args, dtype=None,
             g_idx: Optional[torch.Tensor] = None, global_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    return codec.quantize_tensor(x, scale, zero_point, dtype=dtype, g_idx=g_idx, **_int_args(args, global_scale))


@torch.no_grad()
def dequantize(x_q: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor] = None, args=None, dtype=None,
               g_idx: Optional[torch.Tensor] = None, global_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    if global_scale is not None:
        raise NotImplementedError("global_scale (FP4 tensor-group quantization) is not on the MI355X hot path")
    kw = {}
    if args is not None:
        kw = _int_args(args, None)
        kw.pop("num_bits")
    return codec.dequantize_tensor(x_q, scale, zero_point, dtype=dtype, g_idx=g_idx, **kw)


@torch.no_grad()
def fake_quantize(x: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor], args,
                  g_idx: Optional[torch.Tensor] = None, global_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    return codec.fake_quantize_tensor(x, scale, zero_point, g_idx=g_idx, **_int_args(args, global_scale))
