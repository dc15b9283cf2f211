"""developer script: HBM-cold timing of the W4A16 C-ABI entry points with / without a zero point"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from compressed_tensors_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sets = B.make_sets(dev, 0)
lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
N, G, BF16 = B.N, B.GROUP, _lib.BF16


def mk(zp_mode):
    args = []
    for s in sets:
        zp = s["zp"].data_ptr() if zp_mode else None
        args.append((s["w"].data_ptr(), BF16, s["scale"].data_ptr(), BF16, zp, _lib.I8, N, N, 1, G, N // G, None, 4, BF16, s["packed"].data_ptr(), stream))
    return lambda i: lib.ct_quant_pack(*args[i % len(args)])


def mkd(zp_mode):
    args = []
    for s in sets:
        zp = s["zp"].data_ptr() if zp_mode else None
        args.append((s["packed"].data_ptr(), N, N // 8, N, 4, s["scale"].data_ptr(), BF16, zp, _lib.I8 if zp_mode else -1, 1, G, N // G, None, s["out"].data_ptr(), BF16, stream))
    return lambda i: lib.ct_unpack_dequant(*args[i % len(args)])


one = B.alg_bytes_one_direction()
for rep in range(2):
    for name, fn in (("compress zp=zeros", mk(True)), ("compress zp=None", mk(False)), ("decompress zp=None", mkd(False)), ("decompress zp=zeros", mkd(True))):
        us = B.time_kernel(fn, 64)
        print(f"{name:24s} {us:7.2f} us  {one / us / 1e3:7.1f} GB/s  {one / us / 1e3 / 80:.1f}%")
