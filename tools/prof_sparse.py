"""developer script: run the sparse-bitmask codec a few times (for rocprofv3 --kernel-trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from compressed_tensors_amd import _lib, codec

N = 8192
dev = torch.device("cuda:0")
lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev).manual_seed(7)
items = []
for _ in range(6):
    w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
    w = w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0)
    values, bitmask, ro = codec.bitmask_compress(w)
    items.append(dict(w=w, values=values.clone(), bitmask=bitmask, ro=ro, out=torch.empty_like(w), v2=torch.empty(N * N, dtype=torch.bfloat16, device=dev),
                      bm2=torch.empty_like(bitmask), ro2=torch.empty_like(ro)))
ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
for it in items:
    it["ws"] = torch.empty(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for i in range(24):
    it = items[i % 6]
    lib.ct_bitmask_compress(it["w"].data_ptr(), _lib.BF16, N, N, it["v2"].data_ptr(), N * N, it["bm2"].data_ptr(), it["ro2"].data_ptr(),
                            it["ws"][-1:].data_ptr(), it["ws"].data_ptr(), ws_bytes, stream)
for i in range(24):
    it = items[i % 6]
    lib.ct_bitmask_decompress(it["values"].data_ptr(), it["values"].numel(), it["bitmask"].data_ptr(), it["ro"].data_ptr(), -1, _lib.BF16,
                              N, N, it["out"].data_ptr(), stream)
torch.cuda.synchronize()
print("ok")
