"""Work partitioning (reference distributed/assign.py:12-42): longest-processing-time-first
bin packing — sort items by weight, descending, and give each to the currently lightest bin."""
from typing import Callable, Hashable, TypeVar

__all__ = ["greedy_bin_packing"]

T = TypeVar("T", bound=Hashable)


def greedy_bin_packing(items: list, num_bins: int, item_weight_fn: Callable = lambda x: 1):
    """Returns (items sorted in place by descending weight, bin -> items, item -> bin)."""
    items.sort(key=item_weight_fn, reverse=True)
    bins = [[] for _ in range(num_bins)]
    loads = [0] * num_bins
    where = {}
    for item in items:
        b = loads.index(min(loads))  # first lightest bin, as upstream
        bins[b].append(item)
        where[item] = b
        loads[b] += item_weight_fn(item)
    return items, bins, where
