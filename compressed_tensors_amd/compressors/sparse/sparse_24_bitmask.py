"""sparse-24-bitmask codec (2:4 semi-structured sparsity + bitmask).

Format id and config survive in the reference (config/base.py:18, config/sparse_24_bitmask.py:16-29,
SparsityStructure.TWO_FOUR config/base.py:67); the compressor class does not.  Restated
(SURVEY.md §8a S2): per 4 consecutive elements keep the two largest |x| (the mask has exactly
two bits per quad even when values are zero; ties keep the lower index); `compressed` has shape
(R, C/2); keys `shape`, `compressed`, `bitmask` (no row_offsets: every row holds C/2 values).
"""
from typing import Dict

import torch

from ... import codec
from ...config import CompressionFormat, SparsityStructure
from ..base import BaseCompressor

__all__ = ["Sparse24BitMaskCompressor", "Sparse24BitMaskTensor", "sparse24_bitmask_compress", "sparse24_bitmask_decompress", "get_24_bytemasks"]

COMPRESSION_PARAM_NAMES = ("shape", "compressed", "bitmask")


def get_24_bytemasks(tensor: torch.Tensor) -> torch.Tensor:
    if tensor.numel() % 4 != 0:
        raise ValueError("Tensor size must be a multiple of 4 for TWO_FOUR sparsity")
    return codec.sparse24_mask(tensor)


def sparse24_bitmask_compress(tensor: torch.Tensor, sparsity_structure="2:4"):
    assert len(tensor.shape) == 2, "Only 2D tensors are supported"
    assert SparsityStructure(sparsity_structure) == SparsityStructure.TWO_FOUR, "Only 2:4 sparsity is supported"
    return codec.sparse24_bitmask_compress(tensor)


def sparse24_bitmask_decompress(values: torch.Tensor, bitmasks: torch.Tensor, original_shape) -> torch.Tensor:
    return codec.sparse24_bitmask_decompress(values, bitmasks, original_shape)


class Sparse24BitMaskTensor:
    def __init__(self, shape, compressed: torch.Tensor, bitmask: torch.Tensor):
        self.shape = list(int(s) for s in shape)
        self.compressed = compressed
        self.bitmask = bitmask

    @staticmethod
    def from_dense(tensor: torch.Tensor, sparsity_structure="2:4") -> "Sparse24BitMaskTensor":
        values, bitmask = sparse24_bitmask_compress(tensor, sparsity_structure)
        return Sparse24BitMaskTensor(shape=tensor.shape, compressed=values, bitmask=bitmask)

    def decompress(self) -> torch.Tensor:
        return sparse24_bitmask_decompress(self.compressed, self.bitmask, self.shape)

    def dict(self, name_prefix: str = "", device: str = None) -> Dict[str, torch.Tensor]:
        pre = name_prefix + "." if name_prefix else ""
        out = {
            pre + "shape": torch.tensor(self.shape, dtype=torch.int64),
            pre + "compressed": self.compressed,
            pre + "bitmask": self.bitmask,
        }
        if device is not None:
            out = {k: v.to(device) for k, v in out.items()}
        return out


@BaseCompressor.register(name=CompressionFormat.sparse_24_bitmask.value)
class Sparse24BitMaskCompressor(BaseCompressor):
    @classmethod
    def compression_param_names(cls, scheme=None) -> tuple:
        return COMPRESSION_PARAM_NAMES

    @classmethod
    def compress(cls, state_dict: dict, scheme=None) -> dict:
        state_dict = state_dict.copy()
        weight = state_dict.pop("weight")
        state_dict.update(Sparse24BitMaskTensor.from_dense(weight).dict())
        return state_dict

    @classmethod
    def decompress(cls, state_dict: dict, scheme=None) -> dict:
        state_dict = state_dict.copy()
        parts = {k: state_dict.pop(k) for k in COMPRESSION_PARAM_NAMES}
        state_dict["weight"] = sparse24_bitmask_decompress(parts["compressed"], parts["bitmask"], parts["shape"].tolist())
        return state_dict

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        return False
