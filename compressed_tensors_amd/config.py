"""Wire-format identifiers (reference config/base.py:15-27, :30-82).  The string values appear
in checkpoints' config.json and must match the reference exactly."""
from enum import Enum, unique

__all__ = ["CompressionFormat", "SparsityStructure"]


@unique
class CompressionFormat(str, Enum):
    dense = "dense"
    sparse_bitmask = "sparse-bitmask"
    sparse_24_bitmask = "sparse-24-bitmask"
    int_quantized = "int-quantized"
    float_quantized = "float-quantized"
    naive_quantized = "naive-quantized"
    pack_quantized = "pack-quantized"
    marlin_24 = "marlin-24"
    mixed_precision = "mixed-precision"
    nvfp4_pack_quantized = "nvfp4-pack-quantized"
    mxfp4_pack_quantized = "mxfp4-pack-quantized"
    mxfp8_quantized = "mxfp8-quantized"


@unique
class SparsityStructure(Enum):
    TWO_FOUR = "2:4"
    UNSTRUCTURED = "unstructured"
    ZERO_ZERO = "0:0"

    def __new__(cls, value):
        obj = object.__new__(cls)
        obj._value_ = value.lower() if value is not None else value
        return obj

    @classmethod
    def _missing_(cls, value):
        if value is None:
            return cls.UNSTRUCTURED
        for member in cls:
            if member.value == str(value).lower():
                return member
        raise ValueError(f"{value} is not a valid {cls.__name__}")
