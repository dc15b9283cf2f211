"""Sparse helpers with the reference's names (utils/helpers.py:87-109, 241-343), running on the
GPU: pack_bitmasks / unpack_bitmasks are HIP kernels, the shard helpers are views/copies."""
import torch

from .. import codec

__all__ = ["pack_bitmasks", "unpack_bitmasks", "tensor_follows_mask_structure", "shard_tensor", "combine_shards"]

pack_bitmasks = codec.pack_bitmasks
unpack_bitmasks = codec.unpack_bitmasks


def tensor_follows_mask_structure(tensor: torch.Tensor, mask: str = "2:4") -> bool:
    """utils/helpers.py:87-109: at least n zeros in every chunk of m, else ValueError"""
    n, m = (int(v) for v in mask.split(":"))
    zero_counts = (tensor.reshape(-1, m) == 0).sum(dim=1)
    if not bool(torch.all(zero_counts >= n)):
        raise ValueError()
    return True


def shard_tensor(tensor: torch.Tensor, shard_sizes, dim: int = 0):
    """utils/helpers.py:241-269"""
    if sum(shard_sizes) != tensor.size(dim):
        raise ValueError("Sum of shard_sizes must equal the size of the tensor along the specified dimension.")
    shards, start = [], 0
    for size in shard_sizes:
        shards.append(tensor.narrow(dim, start, size))
        start += size
    return shards


def combine_shards(shards, dim: int = 0) -> torch.Tensor:
    """utils/helpers.py:272-303"""
    if not shards:
        raise ValueError("The list of shards is empty.")
    if len({s.dtype for s in shards}) > 1:
        raise ValueError("All shards must have the same dtype.")
    shape = list(shards[0].shape)
    shape[dim] = sum(s.shape[dim] for s in shards)
    combined = torch.zeros(shape, dtype=shards[0].dtype, device=shards[0].device)
    off = 0
    for s in shards:
        combined.narrow(dim, off, s.shape[dim]).copy_(s)
        off += s.shape[dim]
    return combined
