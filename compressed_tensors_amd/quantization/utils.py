"""Quantization parameter generation on the GPU.

`calculate_qparams_from_weight` fuses the min-max observer with the reference's
calculate_qparams (quantization/utils/helpers.py:50-137) in one streaming pass over the weight.
"""
import torch

from .. import codec
from .quant_args import enum_value

__all__ = ["calculate_qparams_from_weight", "is_module_quantized"]


@torch.no_grad()
def calculate_qparams_from_weight(weight: torch.Tensor, args):
    """scale / zero-point of a 2-D weight for tensor, channel or group strategies.
    Shapes follow the reference's initialisation: tensor -> (1,), channel -> (R, 1), group -> (R, G)."""
    st = enum_value(args.strategy)
    if enum_value(getattr(args, "type", "int")) != "int":
        raise NotImplementedError("only INT quantization is implemented by the MI355X hot path")
    kw = dict(num_bits=int(args.num_bits), symmetric=bool(args.symmetric))
    if st == "tensor":
        scale, zp = codec.minmax_qparams(weight.reshape(1, -1), group_size=None, **kw)
        return scale.reshape(1), zp.reshape(1)
    if st == "channel":
        return codec.minmax_qparams(weight, group_size=None, **kw)
    if st == "group":
        return codec.minmax_qparams(weight, group_size=int(args.group_size), **kw)
    raise NotImplementedError(f"calculate_qparams_from_weight: strategy {st!r} not supported")


def is_module_quantized(module) -> bool:
    """quantization/utils/helpers.py:229-250: a module is quantized when it carries a scheme
    with at least one of weights / input / output args"""
    scheme = getattr(module, "quantization_scheme", None)
    if scheme is None:
        return False
    return any(getattr(scheme, k, None) is not None for k in ("weights", "input_activations", "output_activations"))
