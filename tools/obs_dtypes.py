"""dev helper: min-max observer (+ calculate_qparams) API time per weight dtype / scheme at 8192^2"""
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
for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16), ("fp32", torch.float32)):
    w = torch.randn(N, N, device=dev, generator=g).to(dt)
    for sch, kw in (("g128_sym", dict(num_bits=4, group_size=128, symmetric=True)), ("g128_asym", dict(num_bits=4, group_size=128, symmetric=False)),
                    ("channel8_sym", dict(num_bits=8, group_size=None, symmetric=True))):
        try:
            out[f"{name}_{sch}"] = round(t(lambda: codec.minmax_qparams(w, **kw)), 1)
        except Exception as e:
            out[f"{name}_{sch}"] = repr(e)[:80]
    del w
    torch.cuda.empty_cache()
print(json.dumps(out))
