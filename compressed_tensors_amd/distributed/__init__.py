from .assign import greedy_bin_packing
from .shard import init_dist, is_distributed, module_size, rank_and_world, shard_items, shard_modules, shard_rows

__all__ = [
    "greedy_bin_packing",
    "init_dist",
    "is_distributed",
    "rank_and_world",
    "module_size",
    "shard_items",
    "shard_modules",
    "shard_rows",
]
