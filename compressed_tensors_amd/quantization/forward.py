"""quantize / dequantize / fake_quantize with the reference's signatures
(quantization/lifecycle/forward.py:36-181), executed by the HIP kernels.

Scope: INT quantization (num_bits 1..8), FLOAT 8-bit (float8_e4m3fn) and FLOAT 4-bit (E2M1, optionally under a
`global_scale`: the tensor_group strategy), strategies tensor / channel / token / group / tensor_group / block,
optional activation ordering (g_idx).
"""
from typing import Optional

import torch

from .. import codec
from .quant_args import enum_value

__all__ = ["quantize", "dequantize", "fake_quantize", "calculate_range"]


def _int_args(args, global_scale):
    return dict(
        global_scale=global_scale,
        qtype=enum_value(getattr(args, "type", "int")),
        num_bits=int(args.num_bits),
        strategy=enum_value(args.strategy),
        group_size=getattr(args, "group_size", None),
        block_structure=getattr(args, "block_structure", None),
    )


def calculate_range(quantization_args, device=None):
    """quantization/utils/helpers.py:198-226, as python floats (the kernels take immediates; no device
    tensors are created)."""
    if enum_value(getattr(quantization_args, "type", "int")) == "float":
        if quantization_args.num_bits == 8:
            return -448.0, 448.0
        if quantization_args.num_bits == 4:
            return -6.0, 6.0
        raise NotImplementedError("Range calculation only supported for 4 and 8 bits")
    bit_range = 2.0 ** quantization_args.num_bits
    return -bit_range / 2, bit_range / 2 - 1


@torch.no_grad()
def quantize(x: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor], args, dtype=None,
             g_idx: Optional[torch.Tensor] = None, global_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    return codec.quantize_tensor(x, scale, zero_point, dtype=dtype, g_idx=g_idx, **_int_args(args, global_scale))


@torch.no_grad()
def dequantize(x_q: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor] = None, args=None, dtype=None,
               g_idx: Optional[torch.Tensor] = None, global_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    kw = {"global_scale": global_scale}
    if args is not None:
        kw = _int_args(args, global_scale)
        kw.pop("num_bits")
        kw.pop("qtype")
    return codec.dequantize_tensor(x_q, scale, zero_point, dtype=dtype, g_idx=g_idx, **kw)


@torch.no_grad()
def fake_quantize(x: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor], args,
                  g_idx: Optional[torch.Tensor] = None, global_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    return codec.fake_quantize_tensor(x, scale, zero_point, g_idx=g_idx, **_int_args(args, global_scale))
