"""Module-parallel apply with an optional recouple (reference distributed/module_parallel.py:23-90; SURVEY.md
§8f N3): modules are split over the ranks with the reference's LPT rule, each rank applies the function to
its own share on its own MI355X, and — only when `recouple=True` — the results are replicated so that every
rank ends with the full model, which is what the reference does by default.

The reference recouples with one pickled `broadcast_object_list` per module.  Over RCCL on xGMI that is the
worst shape: many small messages, each staged through host pickling.  Here a rank describes its modules once
(`all_gather_object` of names / shapes / dtypes: a few KB) and then sends ALL its tensors as ONE flat byte
buffer per owner rank (`dist.broadcast`, device to device): `world_size` large collectives in total, and the
receivers' parameters are views into the received buffer (no unpacking copy).  FP8 payloads need no special
case (upstream's `as_broadcastable`, distributed/utils.py:96-110): everything travels as bytes."""
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from ..utils.module import get_direct_state_dict, replace_direct_state_dict
from .assign import greedy_bin_packing
from .shard import is_distributed, module_size, rank_and_world

__all__ = ["replace_module_parallel", "recouple_modules"]

_ALIGN = 256  # every tensor starts on a 256-byte boundary of the flat buffer
_INLINE_NUMEL = 64  # host tensors up to this size travel inside the (pickled) description


def _pad(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


def recouple_modules(modules: Sequence[torch.nn.Module], owner: Dict[torch.nn.Module, int], device: Optional[torch.device] = None) -> None:
    """make every rank hold the direct state dict (and quantization_status) that the OWNER of each module has"""
    rank, world = rank_and_world()
    if world == 1:
        return
    modules = list(modules)
    index = {m: i for i, m in enumerate(modules)}
    mine = [m for m in modules if owner[m] == rank]
    if device is None:
        device = next((t.device for m in mine for t in get_direct_state_dict(m).values() if t is not None and t.device.type != "cpu"), None)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")

    # 1. describe: per owned module, (key, shape, dtype, inline tensor or byte offset into this rank's flat buffer)
    desc, chunks, offset = {}, [], 0
    for m in mine:
        entries = []
        for key, t in get_direct_state_dict(m).items():
            if t is None:
                entries.append((key, None, None, None, None))
            elif t.device.type == "cpu" and t.numel() <= _INLINE_NUMEL:
                entries.append((key, tuple(t.shape), t.dtype, "inline", t.clone()))
            else:
                nbytes = t.numel() * t.element_size()
                entries.append((key, tuple(t.shape), t.dtype, t.device.type, offset))
                chunks.append((offset, t))
                offset += _pad(nbytes)
        desc[index[m]] = (entries, getattr(m, "quantization_status", None))
    gathered = [None] * world
    dist.all_gather_object(gathered, (desc, offset))

    # 2. one broadcast of one flat byte buffer per owner rank
    for src in range(world):
        src_desc, nbytes = gathered[src]
        if nbytes == 0 and not src_desc:
            continue
        flat = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        if src == rank:
            for off, t in chunks:
                n = t.numel() * t.element_size()
                flat[off:off + n].copy_(t.contiguous().reshape(-1).view(torch.uint8), non_blocking=True)
        if nbytes:
            dist.broadcast(flat, src=src)
        if src == rank:
            continue
        for mi, (entries, status) in src_desc.items():
            state = {}
            for key, shape, dtype, where, payload in entries:
                if shape is None:
                    state[key] = None
                elif where == "inline":
                    state[key] = payload
                else:
                    n = 1
                    for s in shape:
                        n *= s
                    n *= torch.empty(0, dtype=dtype).element_size()
                    view = flat[payload:payload + n].view(dtype).view(shape)  # a view into the received buffer
                    state[key] = view if where != "cpu" or device.type == "cpu" else view.cpu()
            m = modules[mi]
            replace_direct_state_dict(m, {k: v for k, v in state.items() if v is not None})
            if status is not None:
                m.quantization_status = status


def replace_module_parallel(modules: List[torch.nn.Module], apply_many_fn: Callable[[List[torch.nn.Module]], None],
                            weight_fn: Callable = module_size, recouple: bool = True) -> List[torch.nn.Module]:
    """Split `modules` over the ranks (LPT by `weight_fn`), apply `apply_many_fn` to this rank's share (a list, so
    that a codec can batch its launches) and, if `recouple`, replicate the results on every rank.  Returns this
    rank's share.  Without torch.distributed it simply applies the function to everything."""
    modules = list(modules)
    if not is_distributed():
        apply_many_fn(modules)
        return modules
    rank, world = rank_and_world()
    order = list(modules)
    if recouple and world > 1:
        # ownership must be identical on every rank even if the replicas have drifted apart (e.g. after a
        # shard-per-rank compress the byte sizes differ per rank): rank 0's sizes decide.  A few hundred ints.
        box = [[int(weight_fn(m)) for m in order]]
        dist.broadcast_object_list(box, src=0)
        size_of = dict(zip(order, box[0]))
        weight_fn = size_of.__getitem__
    _, bins, owner = greedy_bin_packing(order, world, weight_fn)
    mine = bins[rank]
    apply_many_fn(mine)
    if recouple:
        recouple_modules(modules, owner)
    return mine
