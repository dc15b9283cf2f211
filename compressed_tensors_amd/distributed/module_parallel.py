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


def recouple_modules(modules: Sequence[torch.nn.Module], owner: Dict[torch.nn.Module, int], device: Optional[torch.device] = None,
                     names: Optional[Dict[torch.nn.Module, str]] = None) -> None:
    """make every rank hold the direct state dict (and quantization_status) that the OWNER of each module has.  The state
    travels keyed by module NAME (`names`; default: the position in `modules`), never by a list index that the sender and
    the receiver could disagree on: a name the receiver does not know raises."""
    rank, world = rank_and_world()
    if world == 1:
        return
    modules = list(modules)
    if names is None:
        names = {m: str(i) for i, m in enumerate(modules)}
    by_name = {names[m]: m for m in modules}
    if len(by_name) != len(modules):
        raise ValueError("recouple_modules: module names must be unique")
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
        desc[names[m]] = (entries, getattr(m, "quantization_status", None))
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
        for name, (entries, status) in src_desc.items():
            if name not in by_name:
                raise RuntimeError(f"recouple: rank {src} sent the state of module {name!r}, which rank {rank} does not have in its list")
            state = {}
            for key, shape, dtype, where, payload in entries:
                if shape is None:
                    state[key] = None
                elif where == "inline":
                    state[key] = payload
                else:
                    n = 1
                    for s_ in shape:
                        n *= s_
                    n *= torch.empty(0, dtype=dtype).element_size()
                    view = flat[payload:payload + n].view(dtype).view(shape)  # a view into the received buffer
                    state[key] = view if where != "cpu" or device.type == "cpu" else view.cpu()
            m = by_name[name]
            replace_direct_state_dict(m, {k: v for k, v in state.items() if v is not None})
            if status is not None:
                m.quantization_status = status


def replace_module_parallel(modules: List[torch.nn.Module], apply_many_fn: Callable[[List[torch.nn.Module]], None],
                            weight_fn: Callable = module_size, recouple: bool = True, names: Optional[Sequence[str]] = None,
                            done_fn: Optional[Callable[[torch.nn.Module], bool]] = None) -> List[torch.nn.Module]:
    """Split `modules` over the ranks (LPT by `weight_fn`), apply `apply_many_fn` to this rank's share (a list, so
    that a codec can batch its launches) and, if `recouple`, replicate the results on every rank.  Returns this
    rank's share.  Without torch.distributed it simply applies the function to everything that is not done.

    `modules` must be the SAME list on every rank (the same model, e.g. every quantized module in `named_modules` order),
    `names` its module names.  `done_fn(m)` says that THIS rank already holds the result for `m` (`skip_compressed`); the
    ranks may disagree about it — after a shard-per-rank compress every rank has compressed a different subset.

    recouple=True: the ranks first agree on identity (one `all_gather_object` of names / done flags / sizes — a few KB; a
    differing name list raises instead of silently pairing one rank's module 3 with another rank's module 5): a module
    that some rank already holds done is OWNED by the lowest such rank and is not processed again; the rest is LPT-packed
    with rank 0's sizes; modules every rank holds are not sent at all.
    recouple=False (collective-free): the bins come from the unfiltered list, so every rank computes the same assignment
    without talking, and `done_fn` only removes work from a rank's own bin.  When `done_fn` is given the weights must not
    depend on what a rank has already compressed: pass a `weight_fn` that is invariant under compression (ModelCompressor
    passes `dense_numel`)."""
    modules = list(modules)
    done = [bool(done_fn(m)) for m in modules] if done_fn is not None else [False] * len(modules)
    if not is_distributed():
        todo = [m for m, d in zip(modules, done) if not d]
        apply_many_fn(todo)
        return todo
    rank, world = rank_and_world()
    if not recouple or world == 1:
        skip = {id(m) for m, d in zip(modules, done) if d}
        _, bins, _ = greedy_bin_packing(list(modules), world, weight_fn)  # (sorts its argument in place)
        mine = [m for m in bins[rank] if id(m) not in skip]
        apply_many_fn(mine)
        return mine
    names = [str(i) for i in range(len(modules))] if names is None else list(names)
    if len(names) != len(modules) or len(set(names)) != len(names):
        raise ValueError("replace_module_parallel: `names` must be unique and match `modules`")
    gathered = [None] * world
    dist.all_gather_object(gathered, (names, done, [int(weight_fn(m)) for m in modules]))
    for r, (other, _, _) in enumerate(gathered):
        if other != names:
            raise RuntimeError(f"replace_module_parallel: rank {r} lists different modules than rank {rank} "
                               f"({len(other)} vs {len(names)} entries); pass the same module list on every rank")
    size_of = dict(zip(modules, gathered[0][2]))  # rank 0's sizes decide: replicas may have drifted apart
    owner, todo = {}, []
    for i, m in enumerate(modules):
        holders = [r for r in range(world) if gathered[r][1][i]]
        if holders:
            owner[m] = holders[0]  # a rank that already holds the result owns it; nobody recomputes it
        else:
            todo.append(m)
    _, bins, packed = greedy_bin_packing(list(todo), world, size_of.__getitem__)
    owner.update(packed)
    mine = bins[rank]
    apply_many_fn(mine)
    need = [m for i, m in enumerate(modules) if not all(gathered[r][1][i] for r in range(world))]
    if need:
        recouple_modules(need, owner, names={m: names[i] for i, m in enumerate(modules)})
    return mine
