"""Per-module state-dict glue (reference utils/module.py:13-65): raw tensors out of a module's
own parameters/buffers, and back in as non-trainable Parameters."""
from itertools import chain

import torch

__all__ = ["get_direct_state_dict", "replace_direct_state_dict"]


def get_direct_state_dict(module: torch.nn.Module) -> dict:
    out = {}
    for name, t in chain(module._parameters.items(), module._buffers.items()):
        if t is None:
            out[name] = None
        else:
            out[name] = t.data if isinstance(t, (torch.nn.Parameter, getattr(torch.nn, "Buffer", ()))) else t
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
