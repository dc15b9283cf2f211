"""developer script: marlin-24 compress a few times (for rocprofv3 --kernel-trace)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import compressed_tensors_amd as cta
from compressed_tensors_amd import codec

N = 8192
dev = torch.device("cuda:0")
args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
g = torch.Generator(device=dev).manual_seed(13)
w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
w = w * codec.sparse24_mask(w).to(w.dtype)
scale, zp = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
cta.Marlin24Compressor.compress(sd, scheme)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    cta.Marlin24Compressor.compress(sd, scheme)
torch.cuda.synchronize()
print("ms per compress", (time.perf_counter() - t0) / 5 * 1e3)
