"""developer script: HBM-cold timing of the sparse-bitmask decompress (8192^2 bf16, 50 %)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from compressed_tensors_amd import _lib, codec

dev = torch.device("cuda:0")
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
N = 8192
g = torch.Generator(device=dev).manual_seed(7)
items = []
for _ in range(8):
    w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
    w = w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0)
    v, bm, ro = codec.bitmask_compress(w)
    items.append((w, v.clone(), bm, ro, torch.empty_like(w)))
def dec(i):
    w, v, bm, ro, out = items[i % 8]
    lib.ct_bitmask_decompress(v.data_ptr(), v.numel(), bm.data_ptr(), ro.data_ptr(), -1, _lib.BF16, N, N, out.data_ptr(), st)
nnz = items[0][1].numel()
alg = 2 * N * N + 2 * nnz + N * N // 8 + 8 * N
for rep in range(3):
    us = B.time_kernel(dec, 32)
    print(f"bitmask decompress: {us:.2f} us  {alg / us / 1e3:.1f} GB/s  {alg / us / 1e3 / 80:.1f}%")
ok = all(torch.equal(it[4].view(torch.int16), it[0].view(torch.int16)) for it in items)
print("round trip exact:", ok)
