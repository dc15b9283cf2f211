"""pack_to_int32 / unpack_from_int32 under the reference's module path
(compressors/pack_quantized/helpers.py:20-180); the bodies are HIP kernels (csrc/ct_pack.hip)."""
from ...codec import pack_to_int32, unpack_from_int32

__all__ = ["pack_to_int32", "unpack_from_int32"]
