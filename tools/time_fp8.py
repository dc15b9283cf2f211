"""dev timer: FP8 (float-quantized) quantize / dequantize at 8192^2 bf16 channel scales, HBM-cold rotation"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compressed_tensors_amd import codec

dev = torch.device("cuda:0")
N, NSETS = 8192, 16
for strategy, gs, sshape in (("channel", None, (N, 1)), ("group", 32, (N, N // 32))):
    xs = [torch.randn((N, N), device=dev, dtype=torch.bfloat16) for _ in range(NSETS)]
    ss = [(torch.rand(sshape, device=dev) * 0.01 + 0.005).to(torch.bfloat16) for _ in range(NSETS)]
    kw = dict(num_bits=8, strategy=strategy, group_size=gs, qtype="float")
    qs = [codec.quantize_tensor(x, s, None, dtype=torch.float8_e4m3fn, **kw) for x, s in zip(xs, ss)]
    for name, fn in (("quantize", lambda i: codec.quantize_tensor(xs[i], ss[i], None, dtype=torch.float8_e4m3fn, **kw)),
                     ("dequantize", lambda i: codec.dequantize_tensor(qs[i], ss[i], None, strategy=strategy, group_size=gs))):
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
        byts = N * N * 3 + ss[0].numel() * 2
        print(f"fp8 {strategy} {name}: {us:.1f} us  {byts / us / 1e3:.0f} GB/s (incl. python launch overhead)")
