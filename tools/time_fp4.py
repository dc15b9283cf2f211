"""dev timer: FP4 codecs at 8192^2 bf16, HBM-cold rotation (python tools/time_fp4.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compressed_tensors_amd import codec

dev = torch.device("cuda:0")
N, NSETS = 8192, 16
for fmt, group in (("nvfp4", 16), ("mxfp4", 32)):
    xs = [torch.randn((N, N), device=dev, dtype=torch.bfloat16) for _ in range(NSETS)]
    if fmt == "nvfp4":
        ss = [(torch.rand((N, N // group), device=dev) * 2 + 0.1).to(torch.float8_e4m3fn).float() for _ in range(NSETS)]
        gs = torch.tensor([3.7], device=dev)
        kind = "f8e4m3"
    else:
        ss = [torch.full((N, N // group), 0.5, device=dev, dtype=torch.bfloat16) for _ in range(NSETS)]
        gs, kind = None, "e8m0"
    ps = [codec.fp4_quantize_and_pack(x, s, gs, group_size=group) for x, s in zip(xs, ss)]
    cs = [s.to(torch.float8_e4m3fn) if fmt == "nvfp4" else torch.full(s.shape, 126, dtype=torch.uint8, device=dev) for s in ss]
    for name, fn in (("compress", lambda i: codec.fp4_quantize_and_pack(xs[i], ss[i], gs, group_size=group)),
                     ("decompress", lambda i: codec.fp4_unpack_and_dequantize(ps[i], cs[i], gs, group_size=group, scale_kind=kind))):
        for i in range(NSETS):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 4
        e0.record()
        for _ in range(reps):
            for i in range(NSETS):
                fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * NSETS)
        sb = ss[0].numel() * (ss[0].element_size() if name == "compress" else 1)
        byts = N * N * 2.5 + sb
        print(f"{fmt} {name}: {us:.1f} us  {byts / us / 1e3:.0f} GB/s (incl. python launch overhead)")
