"""sparse-bitmask codec.

The reference snapshot keeps the format id ("sparse-bitmask", config/base.py:17), its config
(config/sparse_bitmask.py:12-25) and the bit-order primitives (utils/helpers.py:306-343) but no
longer ships the compressor class (compressors/base.py:43-44 calls it legacy).  This module
restates the historical behaviour on top of the HIP kernels:

    compress(W):   mask = W != 0; compressed = W[mask]; bitmask = packbits_le(mask);
                   row_offsets = exclusive cumsum of the per-row counts; shape = W.shape
    decompress:    zeros(shape)[mask] = compressed

State-dict keys per parameter prefix: `<prefix>.shape`, `.compressed`, `.bitmask`, `.row_offsets`
(prefix = parameter name without `.weight`).  -0.0 counts as zero (comes back +0.0); NaN is kept.
"""
from typing import Dict

import torch

from ... import codec
from ...config import CompressionFormat
from ..base import BaseCompressor

__all__ = ["BitmaskCompressor", "BitmaskTensor", "bitmask_compress", "bitmask_decompress"]

COMPRESSION_PARAM_NAMES = ("shape", "compressed", "bitmask", "row_offsets")


def bitmask_compress(tensor: torch.Tensor, exact: bool = True):
    """-> (values, bitmask, row_offsets); `exact`: see codec.bitmask_compress (False: `values` is a view of a dense-sized buffer)"""
    return codec.bitmask_compress(tensor, exact=exact)


def bitmask_decompress(values: torch.Tensor, bitmasks: torch.Tensor, original_shape, row_offsets=None) -> torch.Tensor:
    return codec.bitmask_decompress(values, bitmasks, original_shape, row_offsets=row_offsets)


class BitmaskTensor:
    """owner of the four tensors of one compressed parameter"""

    def __init__(self, shape, compressed: torch.Tensor, bitmask: torch.Tensor, row_offsets: torch.Tensor):
        self.shape = list(int(s) for s in shape)
        self.compressed = compressed
        self.bitmask = bitmask
        self.row_offsets = row_offsets

    @staticmethod
    def from_dense(tensor: torch.Tensor, exact: bool = True) -> "BitmaskTensor":
        """`exact` (default): `compressed` owns nnz elements, like `tensor[mask]`.  `exact=False` skips the copy that makes it so and hands
        back a view of a dense-sized buffer — faster per call, but the object then holds as much memory as the dense tensor did."""
        shape = tensor.shape
        values, bitmask, row_offsets = bitmask_compress(tensor, exact=exact)
        return BitmaskTensor(shape=shape, compressed=values, bitmask=bitmask, row_offsets=row_offsets)

    @staticmethod
    def from_dense_many(tensors, exact: bool = True) -> "list":
        """`from_dense` for a list of tensors with ONE host wait per window of tensors instead of one per tensor
        (codec.bitmask_compress_many): what a checkpoint of sparse weights should go through"""
        tensors = list(tensors)
        parts = codec.bitmask_compress_many(tensors, exact=exact)
        return [BitmaskTensor(shape=t.shape, compressed=v, bitmask=b, row_offsets=r) for t, (v, b, r) in zip(tensors, parts)]

    def decompress(self) -> torch.Tensor:
        return bitmask_decompress(self.compressed, self.bitmask, self.shape, self.row_offsets)

    def dict(self, name_prefix: str = "", device: str = None) -> Dict[str, torch.Tensor]:
        pre = name_prefix + "." if name_prefix else ""
        out = {
            pre + "shape": torch.tensor(self.shape, dtype=torch.int64),
            pre + "compressed": self.compressed,
            pre + "bitmask": self.bitmask,
            pre + "row_offsets": self.row_offsets,
        }
        if device is not None:
            out = {k: v.to(device) for k, v in out.items()}
        return out


@BaseCompressor.register(name=CompressionFormat.sparse_bitmask.value)
class BitmaskCompressor(BaseCompressor):
    """classmethod interface on local names: {"weight": W} <-> {"shape", "compressed", "bitmask", "row_offsets"}"""

    @classmethod
    def compression_param_names(cls, scheme=None) -> tuple:
        return COMPRESSION_PARAM_NAMES

    @classmethod
    def compress(cls, state_dict: dict, scheme=None) -> dict:
        state_dict = state_dict.copy()
        weight = state_dict.pop("weight")
        state_dict.update(BitmaskTensor.from_dense(weight).dict())
        return state_dict

    @classmethod
    def decompress(cls, state_dict: dict, scheme=None) -> dict:
        state_dict = state_dict.copy()
        parts = {k: state_dict.pop(k) for k in COMPRESSION_PARAM_NAMES if k in state_dict}
        shape = parts["shape"].tolist()
        state_dict["weight"] = bitmask_decompress(parts["compressed"], parts["bitmask"], shape, parts.get("row_offsets"))
        return state_dict

    @classmethod
    def can_compress(cls, module_type: type, scheme) -> bool:
        return False  # never inferred: sparsity formats are chosen explicitly

    # ---- whole-model dictionaries with prefixed names (historical interface)
    @classmethod
    def compress_state_dict(cls, model_state: Dict[str, torch.Tensor], targets=None) -> Dict[str, torch.Tensor]:
        out = {}
        picked = [name for name in model_state if name.endswith(".weight") and (targets is None or name[: -len(".weight")] in targets)]
        # every weight of the checkpoint in one batched pass (one host wait per window of tensors, not one per tensor)
        done = dict(zip(picked, BitmaskTensor.from_dense_many([model_state[name] for name in picked])))
        for name, value in model_state.items():
            if name in done:
                out.update(done[name].dict(name_prefix=name[: -len(".weight")]))
            else:
                out[name] = value
        return out

    @classmethod
    def decompress_state_dict(cls, compressed_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        out, groups = {}, {}
        for name, value in compressed_state.items():
            prefix, _, leaf = name.rpartition(".")
            if leaf in COMPRESSION_PARAM_NAMES:
                groups.setdefault(prefix, {})[leaf] = value
            else:
                out[name] = value
        ready = []
        for prefix, parts in groups.items():
            if not all(k in parts for k in ("shape", "compressed", "bitmask")):
                for leaf, v in parts.items():
                    out[f"{prefix}.{leaf}"] = v
                continue
            out[prefix + ".weight"] = None  # keeps its place in the dictionary's order
            ready.append((prefix, parts))
        # every complete group of the checkpoint in one batched pass (one launch per element size, codec.bitmask_decompress_many)
        dense = codec.bitmask_decompress_many([(p["compressed"], p["bitmask"], p["shape"].tolist(), p.get("row_offsets")) for _, p in ready])
        for (prefix, _), w in zip(ready, dense):
            out[prefix + ".weight"] = w
        return out
