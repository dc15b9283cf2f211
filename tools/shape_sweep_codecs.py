"""real-shape sweep of the other codecs: sparse-bitmask compress / decompress, 2:4 bitmask, int8 / fp8 channel quantize + dequantize, nvfp4 (HBM-cold)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compressed_tensors_amd import _lib, codec
lib = _lib.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
BF16 = _lib.BF16
shapes = [(8192, 8192), (28672, 8192), (8192, 28672), (1024, 8192), (14336, 4096), (4096, 14336), (3584, 3584), (18944, 3584), (3584, 18944),
          (7168, 2048), (2048, 7168), (7168, 18432), (5120, 5120), (13824, 5120), (5120, 13824), (11008, 4096), (4096, 11008)]
which = os.environ.get("WHICH", "sparse,s24,q8").split(",")

def timeit(fn, nsets, n):
    for i in range(nsets + 5): fn(i)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / n)
    return sorted(ts)[1]

def line(name, r, c, alg, uc, ud, ok=True):
    fc, fd = alg / uc / 8e6, alg / ud / 8e6
    flag = "  <<<<" if min(fc, fd) < 0.55 and alg > 6e7 else ""
    print(f"{name:22s} {r:6d}x{c:<6d} {alg/1e6:7.1f} MB: compress {uc:7.1f} us ({fc:.3f})  decompress {ud:7.1f} us ({fd:.3f}) ok={ok}{flag}", flush=True)

for (r, c) in shapes:
    nsets = min(600, max(3, -(-(2 * 256 * 2 ** 20) // (r * c))))  # the kept values / the int8 codes (r x c bytes) >= 2 x the Infinity Cache
    n = max(2 * nsets, 120 if r * c > 3e7 else 200)
    g = torch.Generator(device=dev).manual_seed(5)
    ws = [torch.randn(r, c, device=dev, generator=g, dtype=torch.bfloat16) for _ in range(nsets)]
    if "sparse" in which:
        wsp = [w.masked_fill(torch.rand(r, c, device=dev, generator=g) < 0.5, 0) for w in ws]
        ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(r, c))
        bufs = [(torch.empty(r * c, dtype=torch.bfloat16, device=dev), torch.empty(r, c // 8, dtype=torch.uint8, device=dev), torch.empty(r, dtype=torch.int64, device=dev),
                 torch.empty(ws_bytes // 8 + 1, dtype=torch.int64, device=dev), torch.empty(r, c, dtype=torch.bfloat16, device=dev)) for _ in range(nsets)]
        ca = [(w.data_ptr(), BF16, r, c, v.data_ptr(), v.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(), wk.data_ptr(), ws_bytes, stream) for w, (v, bm, ro, wk, o) in zip(wsp, bufs)]
        uc = timeit(lambda i: lib.ct_bitmask_compress(*ca[i % nsets]), nsets, n)
        nnz = [int(b[3][-1].item()) for b in bufs]
        da = [(v.data_ptr(), nz, bm.data_ptr(), ro.data_ptr(), -1, BF16, r, c, o.data_ptr(), stream) for (v, bm, ro, wk, o), nz in zip(bufs, nnz)]
        ud = timeit(lambda i: lib.ct_bitmask_decompress(*da[i % nsets]), nsets, n)
        ok = torch.equal(bufs[0][4], wsp[0])
        line("sparse-bitmask bf16", r, c, 2 * r * c + 2 * nnz[0] + r * c // 8 + 8 * r, uc, ud, ok)
        del wsp, bufs, ca, da
    if "s24" in which:
        bufs = [(torch.empty(r, c // 2, dtype=torch.bfloat16, device=dev), torch.empty(r, c // 8, dtype=torch.uint8, device=dev), torch.empty(r, c, dtype=torch.bfloat16, device=dev)) for _ in range(nsets)]
        uc = timeit(lambda i: lib.ct_sparse24_compress(ws[i % nsets].data_ptr(), BF16, r, c, bufs[i % nsets][0].data_ptr(), bufs[i % nsets][1].data_ptr(), stream), nsets, n)
        ud = timeit(lambda i: lib.ct_bitmask_decompress(bufs[i % nsets][0].data_ptr(), r * c // 2, bufs[i % nsets][1].data_ptr(), None, c // 2, BF16, r, c, bufs[i % nsets][2].data_ptr(), stream), nsets, n)
        line("sparse-24-bitmask bf16", r, c, 3 * r * c + r * c // 8, uc, ud)
        del bufs
    if "q8" in which:
        sc = [(w.float().abs().amax(dim=1, keepdim=True) / 127.0).to(torch.bfloat16).contiguous() for w in ws]
        qs = [torch.empty(r, c, dtype=torch.int8, device=dev) for _ in range(nsets)]
        outs = [torch.empty(r, c, dtype=torch.bfloat16, device=dev) for _ in range(nsets)]
        uc = timeit(lambda i: lib.ct_quantize(ws[i % nsets].data_ptr(), BF16, sc[i % nsets].data_ptr(), BF16, None, -1, r, c, 1, c, 1, None, 8, BF16, qs[i % nsets].data_ptr(), _lib.I8, stream), nsets, n)
        ud = timeit(lambda i: lib.ct_dequantize(qs[i % nsets].data_ptr(), _lib.I8, sc[i % nsets].data_ptr(), BF16, None, -1, r, c, 1, c, 1, None, outs[i % nsets].data_ptr(), BF16, stream), nsets, n)
        line("int8 channel q / dq", r, c, 3 * r * c + 2 * r, uc, ud)
        del qs, outs, sc
    del ws
    torch.cuda.empty_cache()
