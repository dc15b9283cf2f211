"""developer script: HBM-cold timing of the secondary C-ABI entry points at 8192^2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from compressed_tensors_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
st = torch.cuda.current_stream(dev).cuda_stream
N, G = 8192, 128
BF16, I8, I32 = _lib.BF16, _lib.I8, _lib.I32
K = 8
w = [torch.randn(N, N, dtype=torch.bfloat16, device=dev) for _ in range(K)]
sc = [(x.abs().amax(dim=-1, keepdim=True).float().reshape(N, 1).expand(N, N // G).contiguous() / 7).to(torch.bfloat16) for x in w[:2]]
q = [torch.randint(-8, 8, (N, N), dtype=torch.int8, device=dev) for _ in range(K)]
pk = [torch.empty(N, N // 8, dtype=torch.int32, device=dev) for _ in range(K)]
pk8 = [torch.empty(N, N // 4, dtype=torch.int32, device=dev) for _ in range(K)]
out = [torch.empty(N, N, dtype=torch.bfloat16, device=dev) for _ in range(K)]


def rep(name, fn, nbytes, iters=32):
    us = B.time_kernel(fn, iters)
    print(f"{name:44s} {us:8.2f} us  {nbytes / us / 1e3:7.1f} GB/s  {nbytes / us / 1e3 / 80:5.1f}%")


e = N * N
rep("pack_to_int32 b=4 (int8 -> int32)", lambda i: lib.ct_pack_int32(q[i % K].data_ptr(), N, N, 4, pk[i % K].data_ptr(), N // 8, st), e + e // 2)
rep("unpack_from_int32 b=4", lambda i: lib.ct_unpack_int32(pk[i % K].data_ptr(), N, N // 8, N // 8, N, 4, q[i % K].data_ptr(), st), e + e // 2)
rep("pack_to_int32 b=8", lambda i: lib.ct_pack_int32(q[i % K].data_ptr(), N, N, 8, pk8[i % K].data_ptr(), N // 4, st), 2 * e)
rep("pack_to_int32 b=3", lambda i: lib.ct_pack_int32(q[i % K].data_ptr(), N, N, 3, pk8[i % K].data_ptr(), N * 3 // 32, st), e + e * 3 // 8)
rep("quantize bf16 -> int8 (4 bit, g128)", lambda i: lib.ct_quantize(w[i % K].data_ptr(), BF16, sc[i % 2].data_ptr(), BF16, None, -1, N, N, 1, G, N // G, None, 4, BF16, q[i % K].data_ptr(), I8, st), 3 * e)
rep("dequantize int8 -> bf16 (g128)", lambda i: lib.ct_dequantize(q[i % K].data_ptr(), I8, sc[i % 2].data_ptr(), BF16, None, -1, N, N, 1, G, N // G, None, out[i % K].data_ptr(), BF16, st), 3 * e)
rep("fake_quantize bf16 (4 bit, g128)", lambda i: lib.ct_fake_quantize(w[i % K].data_ptr(), BF16, sc[i % 2].data_ptr(), BF16, None, -1, N, N, 1, G, N // G, None, 4, BF16, out[i % K].data_ptr(), BF16, st), 4 * e)
rep("quant_pack W8 (8 bit, g128) generic g32", lambda i: lib.ct_quant_pack(w[i % K].data_ptr(), BF16, sc[i % 2].data_ptr(), BF16, None, -1, N, N, 1, G, N // G, None, 8, BF16, pk8[i % K].data_ptr(), st), 3 * e)
rep("unpack_dequant W8", lambda i: lib.ct_unpack_dequant(pk8[i % K].data_ptr(), N, N // 4, N, 8, sc[i % 2].data_ptr(), BF16, None, -1, 1, G, N // G, None, out[i % K].data_ptr(), BF16, st), 3 * e)
rep("quant_pack W3 (3 bit, g128) generic g32", lambda i: lib.ct_quant_pack(w[i % K].data_ptr(), BF16, sc[i % 2].data_ptr(), BF16, None, -1, N, N, 1, G, N // G, None, 3, BF16, pk8[i % K].data_ptr(), st), 2 * e + e * 3 // 8)
w32 = [x.float() for x in w[:3]]
sc32 = sc[0].float()
rep("quant_pack W4 fp32 weights (generic)", lambda i: lib.ct_quant_pack(w32[i % 3].data_ptr(), _lib.F32, sc32.data_ptr(), _lib.F32, None, -1, N, N, 1, G, N // G, None, 4, _lib.F32, pk[i % K].data_ptr(), st), 4 * e + e // 2, iters=12)
