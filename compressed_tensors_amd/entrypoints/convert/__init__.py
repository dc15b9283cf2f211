"""Model-free checkpoint conversion (reference entrypoints/convert/): safetensors in,
GPU decompress, safetensors out — the on-disk consumer of the decompress hot path."""
from .convert_checkpoint import convert_checkpoint, convert_file, exec_jobs, validate_file
from .converters import CompressedTensorsDequantizer, Converter, build_inverse_weight_maps

__all__ = ["convert_checkpoint", "convert_file", "validate_file", "exec_jobs", "Converter", "build_inverse_weight_maps",
           "CompressedTensorsDequantizer"]
