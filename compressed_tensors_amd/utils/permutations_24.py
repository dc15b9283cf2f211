"""marlin-24 permutation tables (reference utils/permutations_24.py:20-53).  The HIP packer
computes the weight permutation arithmetically (csrc/ct_marlin24.hip: marlin24_perm_entry); this
module rebuilds the same tables on the host for callers that want them."""
import torch

__all__ = ["get_permutations_24"]


def _weight_perm(num_bits: int):
    if num_bits == 4:
        interleave = (0, 2, 4, 6, 1, 3, 5, 7)
    elif num_bits == 8:
        interleave = (0, 2, 1, 3)
    else:
        raise ValueError("num_bits must be 4 or 8, got {}".format(num_bits))
    flat = []
    for i in range(32):
        col, col_o = i // 4, i // 8
        base = [16 * r + col_o * 256 + 8 * (col % 2) + 4 * blk
                for blk in (0, 1)
                for r in (2 * (i % 4), 2 * (i % 4) + 1, 2 * (i % 4 + 4), 2 * (i % 4 + 4) + 1)]
        for j in range(4):
            flat.extend(p + j for p in base)
    n = len(interleave)
    return [flat[g * n + k] for g in range(len(flat) // n) for k in interleave]


def get_permutations_24(num_bits: int):
    perm = torch.tensor(_weight_perm(num_bits), dtype=torch.int64)
    scale_perm = [i * 8 + j for i in range(8) for j in (0, 4, 1, 5, 2, 6, 3, 7)]
    scale_perm_single = [8 * i + j for i in range(8) for j in range(8)]
    return perm, scale_perm, scale_perm_single
