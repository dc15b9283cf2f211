"""round-3 experiments (dev helper): python tools/exp_r03.py bm   (CT_BITMASK_RESIDENT selects the compress form)
(round 5: the CT_BITMASK_RESIDENT* knobs live in libct_hip_diag.so only — set _lib.LIB_PATH = _lib.DIAG_LIB_PATH before the first load() to use them)"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
what = sys.argv[1]
if what == "bm":
    from compressed_tensors_amd import _lib, codec
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    res = {"mode": os.environ.get("CT_BITMASK_RESIDENT")}
    ok = True
    shapes = ((8192, 8192, 0.5), (4096, 4096, 0.5), (8192, 8192, 0.05), (8192, 8192, 1.0), (1000, 4104, 0.3), (3, 8, 0.5), (257, 2048, 0.0), (12288, 8192, 0.5))
    if len(sys.argv) > 2 and sys.argv[2] == "big":
        shapes = shapes + ((16384, 16384, 0.5), (20000, 16384, 0.7))
    for (r, c, dens) in shapes:
        w = torch.randn(r, c, dtype=torch.bfloat16, device=dev, generator=g)
        if dens < 1.0:
            w = w.masked_fill(torch.rand(r, c, device=dev, generator=g) >= dens, 0)
        v, bm, ro = codec.bitmask_compress(w)
        v2, bm2, ro2 = codec.bitmask_compress(w, two_pass=True)
        same = v.numel() == v2.numel() and torch.equal(v.view(torch.int16), v2.view(torch.int16)) and torch.equal(bm, bm2) and torch.equal(ro, ro2)
        ok = ok and same
        res[f"{r}x{c}@{dens}"] = bool(same)
        del w, v, v2, bm, bm2, ro, ro2
        torch.cuda.empty_cache()
    res["all_equal"] = ok
    for N in (8192, 4096, 2048, 256):
        nsets = 8 if N == 8192 else 6
        ws_ = []
        for i in range(nsets):
            w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
            ws_.append(w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0))
        ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
        wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
        vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev); bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
        f = lambda i: lib.ct_bitmask_compress(ws_[i % nsets].data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(), wk.data_ptr(), ws_bytes, stream)
        f(0); torch.cuda.synchronize()
        import time
        t0 = time.perf_counter(); f(1); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e6
        if one > 1500:  # time-outs inside the kernel: do not spend GPU minutes timing it
            res[f"us_{N}"] = f"single launch {one:.0f} us: skipped"
        else:
            res[f"us_{N}"] = round(B.time_kernel(f, 40), 2)
        res[f"total_{N}"] = int(wk[-1].item())
        del ws_, vals, bm, ro
        torch.cuda.empty_cache()
    print(json.dumps(res))
elif what == "bm32":
    # 32-bit payloads through the sparse-bitmask compress (CT_BITMASK_RESIDENT=0: count / scan / scatter, the round-2 path for them)
    from compressed_tensors_amd import _lib
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    out = {}
    for N, C in ((8192, 8192), (4096, 4096)):
        nsets = 4 if N == 8192 else 12
        ws_ = [torch.randn(N, C, dtype=torch.float32, device=dev, generator=g).masked_fill(torch.rand(N, C, device=dev, generator=g) < 0.5, 0) for _ in range(nsets)]
        ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, C))
        wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
        vals = torch.empty(N * C, dtype=torch.float32, device=dev); bm = torch.empty(N, C // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
        f = lambda i: lib.ct_bitmask_compress(ws_[i % nsets].data_ptr(), _lib.F32, N, C, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(), wk.data_ptr(), ws_bytes, stream)
        us = B.time_kernel(f, 12)
        f(0); torch.cuda.synchronize()
        w0 = ws_[0]; nnz = int(wk[-1].item())
        ok = nnz == int((w0 != 0).sum().item()) and bool(torch.equal(vals[:nnz], w0[w0 != 0]))
        alg = 4 * N * C + 4 * nnz + N * C // 8 + 8 * N
        out[f"{N}x{C}"] = {"us": round(us, 1), "GBps": round(alg / us / 1e3, 1), "frac": round(alg / us / 1e3 / B.HBM_PEAK_GBPS, 4), "exact": ok}
        # decompress of the same payload
        outs = [torch.empty(N, C, dtype=torch.float32, device=dev) for _ in range(nsets)]
        fd = lambda i: lib.ct_bitmask_decompress(vals.data_ptr(), nnz, bm.data_ptr(), ro.data_ptr(), -1, _lib.F32, N, C, outs[i % nsets].data_ptr(), stream)
        usd = B.time_kernel(fd, 12)
        fd(0); torch.cuda.synchronize()
        out[f"{N}x{C}"].update({"decompress_us": round(usd, 1), "decompress_frac": round(alg / usd / 1e3 / B.HBM_PEAK_GBPS, 4), "decompress_exact": bool(torch.equal(outs[0], w0))})
        del ws_, vals, bm, outs; torch.cuda.empty_cache()
    print(json.dumps(out))
elif what == "m24":
    r = B.marlin24_leg(dev)
    print(json.dumps({k: v for k, v in r.items() if "us" in k or "exact" in k}))
elif what == "w4pts":
    out = {}
    for key, kw in (("actorder", dict(n=B.N, actorder=True)), ("actorder_asym", dict(n=B.N, actorder=True, symmetric=False)), ("bf16_4096", dict(n=4096))):
        r = B.w4_kernel_point(dev, **kw)
        out[key] = (r["compress_us"], r["decompress_us"], r["round_trip_equals_fake_quantize"])
        torch.cuda.empty_cache()
    print(json.dumps(out))
elif what == "m24prof":
    import cProfile, pstats, io
    import compressed_tensors_amd as cta
    from compressed_tensors_amd import codec
    N = 8192
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    g = torch.Generator(device=dev).manual_seed(13)
    w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
    w = w * codec.sparse24_mask(w).to(w.dtype)
    scale, zp = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
    sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
    M = cta.Marlin24Compressor
    with M.deferred_structure_check():
        for _ in range(50):
            M.compress(sd, scheme)
    torch.cuda.synchronize()
    import time
    pr = cProfile.Profile()
    with M.deferred_structure_check():
        t0 = time.perf_counter()
        pr.enable()
        for _ in range(400):
            M.compress(sd, scheme)
        pr.disable()
        host = (time.perf_counter() - t0) / 400 * 1e6
    torch.cuda.synchronize()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14)
    print("host us per call (profiled)", round(host, 1)); print(st.getvalue()[:3500])
elif what == "bmstamps":
    from compressed_tensors_amd import _lib
    import numpy as np
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    N = 8192; nsets = 8
    ws_ = [torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g).masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0) for _ in range(nsets)]
    ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
    wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
    vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev); bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
    f = lambda i: lib.ct_bitmask_compress(ws_[i % nsets].data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(), wk.data_ptr(), ws_bytes, stream)
    us = B.time_kernel(f, 40)
    torch.cuda.synchronize()
    st = wk[8196: 8196 + 4 * 512].reshape(512, 4).cpu().double().numpy()
    t0 = st[:, 1].min()
    a = (st - t0) / 100.0
    out = {"us": round(us, 2)}
    for name, sl in (("blocks_0_255", slice(0, 256)), ("blocks_256_511", slice(256, 512))):
        out[name] = {"published": round(float(np.median(a[sl, 1])), 1), "pass2_done": round(float(np.median(a[sl, 0])), 1), "resolved": round(float(np.median(a[sl, 2])), 1), "done": round(float(np.median(a[sl, 3])), 1)}
    print(json.dumps(out))
