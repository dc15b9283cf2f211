from .assign import greedy_bin_packing
from .module_parallel import recouple_modules, replace_module_parallel
from .shard import dense_numel, init_dist, is_distributed, merge_bitmask_row_shards, module_size, rank_and_world, shard_items, shard_modules, shard_rows

__all__ = [
    "greedy_bin_packing",
    "replace_module_parallel",
    "recouple_modules",
    "init_dist",
    "is_distributed",
    "rank_and_world",
    "module_size",
    "dense_numel",
    "shard_items",
    "shard_modules",
    "shard_rows",
    "merge_bitmask_row_shards",
]
