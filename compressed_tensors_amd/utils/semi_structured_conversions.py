"""2:4 semi-structured conversion with the reference's names
(utils/semi_structured_conversions.py:66-330), executed by HIP kernels."""
import torch

from .. import codec

__all__ = ["sparse_semi_structured_from_dense_cutlass", "sparse_semi_structured_to_dense_cutlass", "mask_creator"]

sparse_semi_structured_from_dense_cutlass = codec.cutlass24_from_dense
sparse_semi_structured_to_dense_cutlass = codec.cutlass24_to_dense


def mask_creator(tensor: torch.Tensor) -> torch.Tensor:
    """2:4 magnitude mask as a float tensor of ones/zeros (:301-330)"""
    if tensor.numel() % 4 != 0:
        raise ValueError(f"Tensor of size {tensor.shape} can't be evenly divided into 4 groups")
    return codec.sparse24_mask(tensor.detach()).to(torch.float32)
