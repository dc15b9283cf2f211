"""developer script: what does the per-workgroup table search of the batched W4 kernels cost?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from compressed_tensors_amd import _lib, codec

dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
G = 128


def make(shapes, nsets):
    sets = []
    for _ in range(nsets):
        keep = []
        for r, c in shapes:
            w = torch.randn(r, c, dtype=torch.bfloat16, device=dev)
            s, z = codec.minmax_qparams(w, num_bits=4, group_size=G, symmetric=True)
            keep.append((w, s, z, torch.empty(r, c // 8, dtype=torch.int32, device=dev), torch.empty_like(w)))
        cb = codec.W4Batch([(w, s, z, p, w.shape[0], w.shape[1], G) for (w, s, z, p, o) in keep], "compress", torch.bfloat16)
        db = codec.W4Batch([(p, s, None, o, w.shape[0], w.shape[1], G) for (w, s, z, p, o) in keep], "decompress", torch.bfloat16)
        cb.launch(st)
        sets.append((keep, cb, db))
    return sets


def run(name, shapes, nsets):
    sets = make(shapes, nsets)
    by = sum(2 * r * c + 2 * r * (c // G) + r * c // 2 for r, c in shapes)
    for d, idx in (("compress", 1), ("decompress", 2)):
        us = B.time_kernel(lambda i: sets[i % nsets][idx].launch(st), 3 * nsets)
        print(f"{name:34s} {d:10s} {us:9.2f} us  {by / us / 1e3:7.1f} GB/s  {by / us / 1e3 / 80:5.1f}%")
    del sets
    torch.cuda.empty_cache()


run("1 x 8192x8192", [(8192, 8192)], 16)
run("16 x 2048x2048", [(2048, 2048)] * 16, 16)
run("256 x 512x512", [(512, 512)] * 256, 16)
run("1024 x 256x256", [(256, 256)] * 1024, 16)
tl = [(r, c) for _ in range(22) for (_, r, c) in B.TINYLLAMA_LAYER]
run("TinyLlama 154 modules", tl, 3)
