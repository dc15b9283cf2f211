"""One weight shard per rank, no collectives.

The units of the hot path (modules; or row blocks of one tensor) are independent, so N GPUs
process N disjoint shards and never exchange data.  `init_dist` mirrors the reference's
backend choice (distributed/utils.py:56-85): "nccl" (= RCCL on ROCm) for GPU ranks, "gloo" on
CPU-only hosts (used by the tests).  The only synchronisation offered is a barrier for timing.
"""
import os
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from .assign import greedy_bin_packing

__all__ = ["init_dist", "is_distributed", "rank_and_world", "module_size", "dense_numel", "shard_modules", "shard_rows", "shard_items", "merge_bitmask_row_shards"]


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def rank_and_world():
    if is_distributed():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist(backend: Optional[str] = None) -> None:
    """Join the torchrun-provided process group (env:// rendezvous) and bind this rank to its GPU."""
    if is_distributed():
        return
    for var in ("RANK", "WORLD_SIZE"):
        if var not in os.environ:
            raise ValueError(f"Cannot find distributed environment variable {var}. Launch with torchrun.")
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    use_gpu = torch.cuda.is_available()
    if backend is None:
        backend = "nccl" if use_gpu else "gloo"
    kwargs = {}
    if use_gpu:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend=backend, init_method="env://", **kwargs)


def module_size(module: torch.nn.Module) -> int:
    """bytes held directly by the module (reference offload/utils.py:144-158)"""
    total = 0
    for t in list(module._parameters.values()) + list(module._buffers.values()):
        if t is not None:
            total += t.numel() * t.element_size()
    return total


def dense_numel(module: torch.nn.Module) -> int:
    """number of elements of the module's DENSE weight, whether or not it is currently compressed (`weight_shape` of the packed
    codecs, else the weight / packed tensor's own element count): an LPT weight that every rank computes identically even after
    the ranks have compressed different subsets of the model"""
    shape = getattr(module, "weight_shape", None)
    if shape is not None and shape.numel() >= 1:
        n = 1
        for s in shape.tolist():
            n *= int(s)
        return n
    w = getattr(module, "weight", None)
    if w is not None:
        return int(w.numel())
    return module_size(module)


def shard_items(items: Sequence, weight_fn: Callable = lambda x: 1, rank: Optional[int] = None,
                world_size: Optional[int] = None) -> List:
    """the items this rank owns under LPT bin packing (deterministic on every rank)"""
    r, w = rank_and_world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    indexed = list(range(len(items)))
    _, bins, _ = greedy_bin_packing(indexed, world_size, lambda i: weight_fn(items[i]))
    return [items[i] for i in bins[rank]]


def shard_modules(modules: Sequence[torch.nn.Module], weight_fn: Callable = module_size, rank: Optional[int] = None,
                  world_size: Optional[int] = None) -> List[torch.nn.Module]:
    return shard_items(list(modules), weight_fn, rank, world_size)


def shard_rows(rows: int, rank: Optional[int] = None, world_size: Optional[int] = None, multiple: int = 1):
    """[start, stop) row block of this rank for a single tensor split by rows (rows are
    independent for pack / quantize; bitmask row_offsets are per shard).  Blocks are multiples
    of `multiple` rows except possibly the last."""
    r, w = rank_and_world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    units = (rows + multiple - 1) // multiple
    per, extra = divmod(units, world_size)
    start_u = rank * per + min(rank, extra)
    stop_u = start_u + per + (1 if rank < extra else 0)
    return min(start_u * multiple, rows), min(stop_u * multiple, rows)


def merge_bitmask_row_shards(shards):
    """Reassemble the sparse-bitmask encoding of ONE tensor from its row-block shards, given in row order as
    (values, bitmask, row_offsets) triples with shard-local row offsets: values and bitmask rows concatenate, and a shard's
    offsets are rebased by the number of non-zeros in the shards before it.  Pure bookkeeping (torch.cat on
    whatever device the shards live on); the result equals compressing the whole tensor on one rank."""
    values, bitmasks, offsets, base = [], [], [], 0
    for v, bm, ro in shards:
        values.append(v)
        bitmasks.append(bm)
        offsets.append(ro + base)
        base += v.numel()
    return torch.cat(values), torch.cat(bitmasks), torch.cat(offsets)
