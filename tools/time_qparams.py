"""developer script: HBM-cold timing of ct_minmax_qparams (N1) at 8192^2 bf16 g128"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from compressed_tensors_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
N = 8192
ws = [torch.randn(N, N, dtype=torch.bfloat16, device=dev) for _ in range(8)]
sc = torch.empty(N, N // 128, dtype=torch.bfloat16, device=dev)
zp = torch.empty(N, N // 128, dtype=torch.int8, device=dev)
for sym in (1, 0):
    fn = lambda i: lib.ct_minmax_qparams(ws[i % 8].data_ptr(), _lib.BF16, N, N, 128, 4, sym, sc.data_ptr(), zp.data_ptr(), stream)
    us = B.time_kernel(fn, 48)
    by = 2 * N * N + 3 * N * N // 128
    print(f"minmax_qparams sym={sym}: {us:.2f} us  {by / us / 1e3:.1f} GB/s  {by / us / 1e3 / 80:.1f}%")
