"""round-2 experiments (dev helper): python tools/exp_r02.py w4d|bitmask   (knobs come from the environment)"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
what = sys.argv[1]
if what == "w4d":
    out = {}
    for key, kw in (("bf16_sym", dict(n=B.N)), ("bf16_asym", dict(n=B.N, symmetric=False))):
        r = B.w4_kernel_point(dev, **kw)
        out[key] = (r["compress_us"], r["decompress_us"], r["round_trip_equals_fake_quantize"])
        torch.cuda.empty_cache()
    print(json.dumps({"rowlead": os.environ.get("CT_W4D_ROWLEAD"), **out}))
elif what == "bitmask":
    r = B.bitmask_leg(dev)
    print(json.dumps({"chunk_mb": os.environ.get("CT_BITMASK_CHUNK_MB"), "compress_us": r["compress_us"], "decompress_us": r["decompress_us"], "ok": r["round_trip_bit_exact"]}))
elif what == "rtn8":
    from compressed_tensors_amd import _lib
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    N = B.N; nsets = 10
    g = torch.Generator(device=dev).manual_seed(3)
    ws = [torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g) for _ in range(nsets)]
    q8 = [torch.empty(N, N, dtype=torch.uint8, device=dev) for _ in range(nsets)]
    sc = torch.empty(N, 1, dtype=torch.bfloat16, device=dev); zp = torch.empty(N, 1, dtype=torch.int8, device=dev)
    out = {}
    for name, fp8, sym in (("fp8", 1, 1), ("int8_sym", 0, 1), ("int8_asym", 0, 0)):
        f = lambda i: lib.ct_rtn_quant_channel8(ws[i % nsets].data_ptr(), _lib.BF16, N, N, fp8, sym, q8[i % nsets].data_ptr(), sc.data_ptr(), zp.data_ptr(), stream)
        out[name] = round(B.time_kernel(f, 40), 2)
    print(json.dumps({"wave": os.environ.get("CT_RTN8_WAVE"), **out}))
elif what == "qp":
    r = B.qparams_leg(dev)
    print(json.dumps({"U": os.environ.get("CT_QP_U"), "us": r["us"], "fused_us": r["fused_with_compress"]["us"]}))
elif what == "bm2":
    r = B.bitmask_leg(dev)
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("CT_BITMASK")}, "compress_us": r["compress_us"], "ok": r["round_trip_bit_exact"]}))
elif what == "bmres":
    # resident bitmask compress: run with CT_BITMASK_RESIDENT=1|2|3; parity against a CT_BITMASK_RESIDENT=0 subprocess dump is done by
    # comparing with the count / scan / scatter form (two_pass=True) in-process
    from compressed_tensors_amd import _lib, codec
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    res = {"mode": os.environ.get("CT_BITMASK_RESIDENT")}
    ok = True
    for (r, c, dens) in ((8192, 8192, 0.5), (4096, 4096, 0.5), (8192, 8192, 0.05), (8192, 8192, 1.0), (1000, 4104, 0.3), (3, 8, 0.5), (257, 2048, 0.0), (12288, 8192, 0.5), (16384, 16384, 0.5)):
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
    for N in (8192, 4096, 2048, 1024, 256):
        nsets = 6
        ws_ = []
        for i in range(nsets):
            w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
            ws_.append(w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0))
        ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
        wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
        vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev); bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
        f = lambda i: lib.ct_bitmask_compress(ws_[i % nsets].data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(), wk.data_ptr(), ws_bytes, stream)
        res[f"us_{N}"] = round(B.time_kernel(f, 40), 2)
        res[f"total_{N}"] = int(wk[-1].item())
        if os.environ.get("CT_BITMASK_RESIDENT") == "3":
            torch.cuda.synchronize()
            st = wk[8196: 8196 + 4 * 512].reshape(512, 4).cpu().double()
            st = st[st[:, 3] > 0]
            if st.shape[0] > 256:
                aa = ((st - st[:, 0].min()) / 100.0).numpy()
                import numpy as np
                res[f"rounds_{N}"] = [[round(float(np.median(aa[sl, k])), 1) for k in range(4)] + [round(float(aa[sl, 3].max()), 1)] for sl in (slice(0, 256), slice(256, None))]
            st = (st - st[:, 0].min()) / 100.0
            import numpy as np
            a = st.numpy()
            wk[8196: 8196 + 4 * 512] = 0
            res[f"stamps_{N}"] = {"wgs": int(a.shape[0]), "start_max": float(a[:, 0].max()), "published_med": float(np.median(a[:, 1])), "published_max": float(a[:, 1].max()),
                                  "resolved_med": float(np.median(a[:, 2])), "resolved_max": float(a[:, 2].max()), "done_med": float(np.median(a[:, 3])), "done_max": float(a[:, 3].max())}
        del ws_, vals, bm, ro
        torch.cuda.empty_cache()
    print(json.dumps(res))
elif what == "bmres1":
    # a few launches of the bitmask compress at 8192^2 (for rocprofv3 counter passes)
    from compressed_tensors_amd import _lib
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    N = 8192
    w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
    w = w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0)
    ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
    wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
    vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev); bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
    for _ in range(10):
        lib.ct_bitmask_compress(w.data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(), wk.data_ptr(), ws_bytes, stream)
    torch.cuda.synchronize()
    print(int(wk[-1].item()))
elif what == "m24host":
    # host cost of Marlin24Compressor.compress: a tiny weight (kernel ~ few us), many calls
    import time, cProfile, pstats, io
    import compressed_tensors_amd as cta, oracle as O
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    w = torch.randn(64, 256, dtype=torch.bfloat16); w = w * O.sparse24_mask(w).to(w.dtype)
    s, z = O.calculate_qparams_minmax(w.to(torch.float16), num_bits=4, group_size=128, symmetric=True)
    sd = {"weight": w.to(dev), "weight_scale": s.to(torch.bfloat16).to(dev), "weight_zero_point": z.to(dev)}
    M = cta.Marlin24Compressor
    with M.deferred_structure_check():
        for _ in range(200): M.compress(sd, scheme)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with M.deferred_structure_check():
        for _ in range(2000): M.compress(sd, scheme)
    torch.cuda.synchronize()
    print("us per call (deferred):", (time.perf_counter() - t0) / 2000 * 1e6)
    pr = cProfile.Profile(); pr.enable()
    with M.deferred_structure_check():
        for _ in range(2000): M.compress(sd, scheme)
    pr.disable(); torch.cuda.synchronize()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14); print(st.getvalue()[:2600])
