"""Light-weight mirrors of the reference's quantization config objects.

The reference defines these as pydantic models (quantization/quant_args.py:169-429,
quant_scheme.py:26-120, quant_config.py:56-121).  The hot path only reads a handful of
attributes from them, so the compressors in this package are duck-typed: they accept the
reference's own objects unchanged (drop-in use) or these plain dataclasses (standalone use,
tests).  No arithmetic lives here.
"""
from dataclasses import dataclass, field
from enum import Enum
from typing import List, Optional

import torch

__all__ = [
    "QuantizationType",
    "QuantizationStrategy",
    "ActivationOrdering",
    "QuantizationStatus",
    "QuantizationArgs",
    "QuantizationScheme",
]


class QuantizationType(str, Enum):
    INT = "int"
    FLOAT = "float"


class QuantizationStrategy(str, Enum):
    TENSOR = "tensor"
    CHANNEL = "channel"
    GROUP = "group"
    BLOCK = "block"
    TOKEN = "token"
    TENSOR_GROUP = "tensor_group"
    ATTN_HEAD = "attn_head"


class ActivationOrdering(str, Enum):
    GROUP = "group"
    WEIGHT = "weight"
    DYNAMIC = "dynamic"


class QuantizationStatus(str, Enum):
    """lifecycle stages the compressors set on modules (quant_config.py:56-121)"""

    INITIALIZED = "initialized"
    CALIBRATION = "calibration"
    FROZEN = "frozen"
    COMPRESSED = "compressed"
    DECOMPRESSED = "decompressed"


@dataclass
class QuantizationArgs:
    num_bits: int = 8
    type: QuantizationType = QuantizationType.INT
    symmetric: bool = True
    group_size: Optional[int] = None
    strategy: Optional[QuantizationStrategy] = None
    block_structure: Optional[List[int]] = None
    dynamic: bool = False
    actorder: Optional[ActivationOrdering] = None
    scale_dtype: Optional[torch.dtype] = None  # quant_args.py:201-202
    zp_dtype: Optional[torch.dtype] = None

    def __post_init__(self):
        self.type = QuantizationType(getattr(self.type, "value", self.type))
        if self.actorder is not None:
            self.actorder = ActivationOrdering(getattr(self.actorder, "value", self.actorder))
        # strategy inference as in quant_args.py:290-330
        if self.strategy is None:
            if self.group_size is not None and self.group_size > 0:
                self.strategy = QuantizationStrategy.GROUP
            elif self.group_size == -1:
                self.strategy = QuantizationStrategy.CHANNEL
            else:
                self.strategy = QuantizationStrategy.TENSOR
        self.strategy = QuantizationStrategy(getattr(self.strategy, "value", self.strategy))
        if self.strategy in (QuantizationStrategy.GROUP, QuantizationStrategy.TENSOR_GROUP):
            if self.group_size is None or self.group_size <= 0:
                raise ValueError(f"strategy {self.strategy} requires group_size to be set to a positive value")
        if self.strategy == QuantizationStrategy.BLOCK and self.block_structure is None:
            raise ValueError("strategy block requires block_structure")

    def pytorch_dtype(self) -> torch.dtype:
        """quant_args.py:413-427"""
        if self.type == QuantizationType.FLOAT:
            if self.num_bits == 8:
                return torch.float8_e4m3fn
            raise NotImplementedError("Only num_bits in (8) are supported")
        if self.num_bits <= 8:
            return torch.int8
        if self.num_bits <= 16:
            return torch.int16
        return torch.int32


@dataclass
class QuantizationScheme:
    targets: List[str] = field(default_factory=list)
    weights: Optional[QuantizationArgs] = None
    input_activations: Optional[QuantizationArgs] = None
    output_activations: Optional[QuantizationArgs] = None
    format: Optional[str] = None


def is_scheme(obj) -> bool:
    """duck-type test that accepts the reference's pydantic QuantizationScheme too"""
    return obj is not None and hasattr(obj, "weights") and hasattr(obj, "input_activations")


def enum_value(v):
    return getattr(v, "value", v)
