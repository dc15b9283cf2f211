"""developer script: does running compress and decompress on two HIP streams hide the per-launch ramp/tail?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from compressed_tensors_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
sets = B.make_sets(dev, 0)
lib = _lib.load()
N, G, BF16 = B.N, B.GROUP, _lib.BF16
s0 = torch.cuda.current_stream(dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def launchers(sc, sd):
    ca, da = [], []
    for s in sets:
        ca.append((s["w"].data_ptr(), BF16, s["scale"].data_ptr(), BF16, s["zp"].data_ptr(), _lib.I8, N, N, 1, G, N // G, None, 4, BF16, s["packed"].data_ptr(), sc.cuda_stream))
        da.append((s["packed"].data_ptr(), N, N // 8, N, 4, s["scale"].data_ptr(), BF16, None, -1, 1, G, N // G, None, s["out"].data_ptr(), BF16, sd.cuda_stream))
    return (lambda i: lib.ct_quant_pack(*ca[i % len(ca)])), (lambda i: lib.ct_unpack_dequant(*da[i % len(da)]))


def run(name, c, d, steps=200):
    for i in range(len(sets)):
        c(i)
    torch.cuda.synchronize()
    for i in range(20):
        c(i); d(i + 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        c(i); d(i + 8)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:28s} {dt * 1e6:7.2f} us/step  {2 * B.alg_bytes_one_direction() / dt / 1e9:7.1f} GB/s  {2 * B.alg_bytes_one_direction() / dt / 1e9 / 80:.1f}%")


for rep in range(2):
    c, d = launchers(s0, s0)
    run("one stream", c, d)
    c, d = launchers(s1, s2)
    run("two streams (c | d)", c, d)
