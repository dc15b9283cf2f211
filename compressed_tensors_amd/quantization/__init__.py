from .forward import calculate_range, dequantize, fake_quantize, quantize
from .quant_args import (
    ActivationOrdering,
    QuantizationArgs,
    QuantizationScheme,
    QuantizationStatus,
    QuantizationStrategy,
    QuantizationType,
)
from .utils import calculate_qparams_from_weight, is_module_quantized

__all__ = [
    "quantize",
    "dequantize",
    "fake_quantize",
    "calculate_range",
    "calculate_qparams_from_weight",
    "is_module_quantized",
    "QuantizationArgs",
    "QuantizationScheme",
    "QuantizationStatus",
    "QuantizationStrategy",
    "QuantizationType",
    "ActivationOrdering",
]
