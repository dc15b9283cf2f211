"""dev helper: 2:4 bitmask (S2) compress / decompress API time per dtype at 8192^2"""
import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from compressed_tensors_amd import codec
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
N = 8192
out = {}
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32), ("fp8", torch.float8_e4m3fn)):
    w = torch.randn(N, N, device=dev, generator=g)
    w = w.to(dt) if dt != torch.float8_e4m3fn else w.to(torch.bfloat16).to(dt)
    v, bm = codec.sparse24_bitmask_compress(w)
    es = w.element_size()
    out[name] = {"compress_api_us": round(t(lambda: codec.sparse24_bitmask_compress(w)), 1), "decompress_api_us": round(t(lambda: codec.sparse24_bitmask_decompress(v, bm, w.shape)), 1),
                 "alg_MB": round((N * N * es * 1.5 + N * N / 8) / 1e6, 1)}
    del w, v, bm
    torch.cuda.empty_cache()
print(json.dumps(out))
