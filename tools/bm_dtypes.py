import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench as B
from compressed_tensors_amd import codec
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
out = {}
for name, dt, N in (("fp32", torch.float32, 8192), ("bf16", torch.bfloat16, 8192), ("int8", torch.int8, 8192), ("fp32_4096", torch.float32, 4096)):
    w = torch.randn(N, N, device=dev, generator=g)
    w = w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0)
    w = (w * 50).to(dt) if dt == torch.int8 else w.to(dt)
    v, bm, ro = codec.bitmask_compress(w)
    torch.cuda.synchronize()
    import time
    def t(f, n=10):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    out[name] = {"compress_api_us": round(t(lambda: codec.bitmask_compress(w)), 1), "decompress_api_us": round(t(lambda: codec.bitmask_decompress(v, bm, w.shape, ro)), 1),
                 "alg_MB": round((w.numel() * w.element_size() + v.numel() * v.element_size() + bm.numel()) / 1e6, 1)}
    del w, v, bm, ro
    torch.cuda.empty_cache()
print(json.dumps(out))
