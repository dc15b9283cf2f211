"""Per-module state-dict glue (reference utils/module.py:13-65): raw tensors out of a module's
own parameters/buffers, and back in as non-trainable Parameters.

`swap_direct_entries` is the same replacement expressed as a delta (names to drop, tensors to add) for the batched
module paths: a 154-module checkpoint spends more host time in `delattr` / `setattr` / `nn.Parameter` churn for entries that
do not change than its two kernels run for (bench.py `tinyllama_checkpoint.api`), so the entries that stay are left alone and
the ones that change are written into `module._parameters` directly when nothing observes the difference."""
from itertools import chain

import torch

__all__ = ["get_direct_state_dict", "replace_direct_state_dict", "swap_direct_entries", "direct_entry"]

_BUFFER = getattr(torch.nn, "Buffer", ())
_MODULE_SETATTR = torch.nn.Module.__setattr__
_MODULE_DELATTR = torch.nn.Module.__delattr__


def get_direct_state_dict(module: torch.nn.Module) -> dict:
    out = {}
    for name, t in chain(module._parameters.items(), module._buffers.items()):
        if t is None:
            out[name] = None
        else:
            out[name] = t.data if isinstance(t, (torch.nn.Parameter, _BUFFER)) else t
    return out


def replace_direct_state_dict(module: torch.nn.Module, new_state_dict: dict) -> None:
    old = get_direct_state_dict(module)
    for name in old:
        if name not in new_state_dict:
            delattr(module, name)
    for name, value in new_state_dict.items():
        if name in old:
            if old[name] is value:  # untouched tensors are returned by identity: leave them
                continue
            delattr(module, name)
        setattr(module, name, torch.nn.Parameter(value, requires_grad=False))


def direct_entry(module: torch.nn.Module, name: str):
    """the module's own parameter or buffer `name` as stored (no `.data` unwrapping), or None"""
    t = module._parameters.get(name)
    if t is None:
        t = module._buffers.get(name)
    return t


def _plain_module(module) -> bool:
    """True when writing `module._parameters` directly cannot be told from setattr / delattr: the class keeps nn.Module's own
    attribute hooks and no global parameter-registration hook is installed"""
    cls = type(module)
    if cls.__setattr__ is not _MODULE_SETATTR or cls.__delattr__ is not _MODULE_DELATTR:
        return False
    return not torch.nn.modules.module._global_parameter_registration_hooks


def swap_direct_entries(module: torch.nn.Module, remove, add: dict, status=None) -> None:
    """The effect of `replace_direct_state_dict(module, new)` where `new` is the module's direct state dict without the names in
    `remove` and with the tensors of `add` (reference utils/module.py:33-65): removed names are deleted, added tensors become
    non-trainable Parameters, and every entry that stays ends as a non-trainable Parameter over the same storage — without
    re-creating the ones that already are.  `status`: also set `module.quantization_status`."""
    params, buffers, attrs = module._parameters, module._buffers, module.__dict__
    plain = _plain_module(module)
    if plain:
        for n in add:
            if n in attrs:
                plain = False
    if not plain:
        new = {k: v for k, v in get_direct_state_dict(module).items() if k not in remove}
        new.update(add)
        replace_direct_state_dict(module, new)
        if status is not None:
            module.quantization_status = status
        return
    for name in remove:
        if name in params:
            del params[name]
        elif name in buffers:
            del buffers[name]
            module._non_persistent_buffers_set.discard(name)
    if buffers:
        # upstream's identity test keeps a plain-tensor buffer where it is (the state dict hands the same object back), while a
        # torch.nn.Buffer entry is unwrapped by `.data`, never matches, and is re-registered as a Parameter — like an added name
        for name in list(buffers):
            t = buffers[name]
            if name in add or isinstance(t, _BUFFER):
                del buffers[name]
                module._non_persistent_buffers_set.discard(name)
                if name not in add:
                    params[name] = torch.nn.Parameter(t.data, requires_grad=False)
    for name, t in list(params.items()):
        if t is not None and t.requires_grad and name not in add:
            # a fresh Parameter, as upstream's replace_direct_state_dict makes for every staying entry: freezing the object in place
            # would also freeze it for any other holder (a tied bias, an optimizer's param group)
            # (a plain torch.nn.Parameter whatever subclass the entry was, as upstream's `torch.nn.Parameter(data, requires_grad=False)` and as
            # the non-plain path above, which calls upstream's function: the two paths of this function agree — ADVICE r05)
            params[name] = _make_subclass(_Parameter, t.data, False)
    for name, value in add.items():
        params[name] = _make_subclass(_Parameter, value, False)  # == torch.nn.Parameter(value, requires_grad=False) for a plain tensor
    if status is not None:
        attrs["quantization_status"] = status  # a plain attribute: what nn.Module.__setattr__ ends up doing for it


_Parameter = torch.nn.Parameter
_make_subclass = torch.Tensor._make_subclass
