from .helpers import combine_shards, pack_bitmasks, shard_tensor, tensor_follows_mask_structure, unpack_bitmasks
from .impl_backend import ImplBackend
from .module import get_direct_state_dict, replace_direct_state_dict
from .permutations_24 import get_permutations_24
from .semi_structured_conversions import (
    mask_creator,
    sparse_semi_structured_from_dense_cutlass,
    sparse_semi_structured_to_dense_cutlass,
)

__all__ = [
    "ImplBackend",
    "get_direct_state_dict",
    "replace_direct_state_dict",
    "pack_bitmasks",
    "unpack_bitmasks",
    "tensor_follows_mask_structure",
    "shard_tensor",
    "combine_shards",
    "get_permutations_24",
    "mask_creator",
    "sparse_semi_structured_from_dense_cutlass",
    "sparse_semi_structured_to_dense_cutlass",
    "getattr_chain",
]


def getattr_chain(obj, chain_str: str, *default):
    """dotted getattr with optional default (reference utils/helpers.py getattr_chain)"""
    try:
        for name in chain_str.split("."):
            obj = getattr(obj, name)
        return obj
    except AttributeError:
        if default:
            return default[0]
        raise
