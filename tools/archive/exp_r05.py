"""round-5 experiments (dev helper, run on the GPU box through tools/gpu_run.sh exp:<name>[,args]): one JSON line per call.

bm [n] [dtype]   sparse-bitmask compress, n x n (8192) bf16|f16|f32|i16 at 50 % zeros: HBM-cold time of ct_bitmask_compress (6 rotating inputs,
                 BLOCKS x 60 launches) + parity against the count / scan / scatter form, under whatever CT_BM_X says (read once per process)
marlin           marlin-24 leg of bench.py: kernel time, class call in default / deferred mode
"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
what = sys.argv[1]
from compressed_tensors_amd import _lib, codec

lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream

if what == "bm":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    name = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "i16": torch.int16}[name]
    g = torch.Generator(device=dev).manual_seed(11)
    ws_ = []
    for _ in range(6):
        w = torch.randn(n, n, dtype=torch.float32, device=dev, generator=g)
        w = (w * 100).to(dt) if dt is torch.int16 else w.to(dt)
        ws_.append(w.masked_fill(torch.rand(n, n, device=dev, generator=g) < 0.5, 0))
    code = _lib.DT[dt]
    ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(n, n))
    wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
    vals = torch.empty(n * n, dtype=dt, device=dev)
    bm = torch.empty(n, n // 8, dtype=torch.uint8, device=dev)
    ro = torch.empty(n, dtype=torch.int64, device=dev)
    f = lambda i: lib.ct_bitmask_compress(ws_[i % 6].data_ptr(), code, n, n, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(),
                                          wk.data_ptr(), ws_bytes, stream)
    res = {"CT_BM_X": os.environ.get("CT_BM_X"), "n": n, "dtype": name}
    sp = {}
    res["us"] = round(B.time_kernel(f, 60, spread=sp), 2)
    res.update(sp)
    torch.cuda.synchronize()
    x = ws_[5]  # the last launch's input ((60 - 1) % 6)
    v2, bm2, ro2 = codec.bitmask_compress(x, two_pass=True)
    nnz = int(wk[-1].item())
    iv = torch.int16 if dt.itemsize == 2 else torch.int32
    res["ok"] = bool(nnz == v2.numel() and torch.equal(vals[:nnz].view(iv), v2.view(iv)) and torch.equal(bm, bm2) and torch.equal(ro, ro2))
    alg = n * n * dt.itemsize + nnz * dt.itemsize + n * n // 8 + 8 * n
    res["frac"] = round(alg / res["us"] / 1e3 / 8000.0, 4)
    print(json.dumps(res))
elif what == "marlin":
    r = B.marlin24_leg(dev)
    print(json.dumps({k: r.get(k) for k in ("kernels_us", "kernels_frac_hbm", "compress_us_default", "compress_us_default_min_max", "compress_us_deferred_check",
                                            "host_issue_us_per_call", "bit_exact_vs_oracle")}))
