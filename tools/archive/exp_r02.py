"""round-2 experiments (dev helper): python tools/exp_r02.py w4d|bitmask   (knobs come from the environment)
(round 5: the CT_BITMASK_RESIDENT* knobs live in libct_hip_diag.so only — set _lib.LIB_PATH = _lib.DIAG_LIB_PATH before the first load() to use them)"""
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
    print(json.dumps({**out}))
elif what == "bitmask":
    r = B.bitmask_leg(dev)
    print(json.dumps({"chunk_mb": os.environ.get(""), "compress_us": r["compress_us"], "decompress_us": r["decompress_us"], "ok": r["round_trip_bit_exact"]}))
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
    print(json.dumps({**out}))
elif what == "qp":
    r = B.qparams_leg(dev)
    print(json.dumps({"us": r["us"], "fused_us": r["fused_with_compress"]["us"]}))
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
elif what == "f32":
    # fp32 weights and scales: W4 g128 compress / decompress, int8 channel-wise quantize / dequantize (ct_quant_pack, ct_unpack_dequant, ct_quantize, ct_dequantize)
    from compressed_tensors_amd import _lib, codec
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    N = 8192; nsets = 4
    g = torch.Generator(device=dev).manual_seed(5)
    res = {}
    for sym in (True, False):
        sets = []
        for _ in range(nsets):
            w = torch.randn(N, N, dtype=torch.float32, device=dev, generator=g)
            sc, zp = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=sym)
            sets.append((w, sc, zp, torch.empty(N, N // 8, dtype=torch.int32, device=dev), torch.empty(N, N, dtype=torch.float32, device=dev)))
        F = _lib.DT[torch.float32]
        ca = [(w.data_ptr(), F, sc.data_ptr(), F, zp.data_ptr(), _lib.I8, N, N, 1, 128, N // 128, None, 4, F, pk.data_ptr(), stream) for (w, sc, zp, pk, out) in sets]
        da = [(pk.data_ptr(), N, N // 8, N, 4, sc.data_ptr(), F, None if sym else zp.data_ptr(), -1 if sym else _lib.I8, 1, 128, N // 128, None, out.data_ptr(), F, stream) for (w, sc, zp, pk, out) in sets]
        def c(i): _lib.check(lib.ct_quant_pack(*ca[i % nsets]))
        def d(i): _lib.check(lib.ct_unpack_dequant(*da[i % nsets]))
        for i in range(nsets): c(i)
        # 12 packed inputs (400 MB) and 2 outputs for the decompress timing: the reads cannot come from the 256 MiB Infinity Cache
        pks = [sets[i % nsets][3].clone() for i in range(12)]
        da = [(pks[i].data_ptr(), N, N // 8, N, 4, sets[i % nsets][1].data_ptr(), F, None if sym else sets[i % nsets][2].data_ptr(), -1 if sym else _lib.I8, 1, 128, N // 128, None,
               sets[i % 2][4].data_ptr(), F, stream) for i in range(12)]
        def d(i): _lib.check(lib.ct_unpack_dequant(*da[i % 12]))
        us_c, us_d = B.time_kernel(c, 16), B.time_kernel(d, 24)
        d(0); torch.cuda.synchronize()
        w, sc, zp, pk, out = sets[0]
        ok = torch.equal(out, codec.fake_quantize_tensor(w, sc, zp, num_bits=4, strategy="group", group_size=128))
        alg = N * N * 4 + N * N // 2 + N * (N // 128) * 4
        res["w4_sym" if sym else "w4_asym"] = {"compress_us": round(us_c, 1), "decompress_us": round(us_d, 1), "alg_MB": round(alg / 1e6, 1), "compress_frac": round(alg / us_c / 8e6, 3),
                                               "decompress_frac": round(alg / us_d / 8e6, 3), "round_trip_equals_fake_quantize": bool(ok)}
        del sets, ca, da
        torch.cuda.empty_cache()
    print(json.dumps(res))
elif what == "f32q":
    # int8 channel-wise quantize / dequantize / fake-quantize kernels through the C ABI (symmetric and asymmetric), fp32 and bf16 weights
    from compressed_tensors_amd import _lib
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    N = 8192
    g = torch.Generator(device=dev).manual_seed(5)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        D = _lib.DT[dt]
        ws = [torch.randn(N, N, dtype=torch.float32, device=dev, generator=g).to(dt) for _ in range(4)]
        sc = (ws[0].float().abs().amax(dim=1, keepdim=True) / 127.0).to(dt).contiguous()
        zp = torch.randint(-3, 4, (N, 1), dtype=torch.int8, device=dev)
        qs = [torch.empty(N, N, dtype=torch.int8, device=dev) for _ in range(4)]
        outs = [torch.empty(N, N, dtype=dt, device=dev) for _ in range(2)]
        es = ws[0].element_size()
        for name, z in (("sym", None), ("asym", zp)):
            zptr = None if z is None else z.data_ptr()
            zdt = -1 if z is None else _lib.I8
            fq = lambda i: _lib.check(lib.ct_quantize(ws[i % 4].data_ptr(), D, sc.data_ptr(), D, zptr, zdt, N, N, 1, N, 1, None, 8, D, qs[i % 4].data_ptr(), _lib.I8, stream))
            fd = lambda i: _lib.check(lib.ct_dequantize(qs[i % 4].data_ptr(), _lib.I8, sc.data_ptr(), D, zptr, zdt, N, N, 1, N, 1, None, outs[i % 2].data_ptr(), D, stream))
            ff = lambda i: _lib.check(lib.ct_fake_quantize(ws[i % 4].data_ptr(), D, sc.data_ptr(), D, zptr, zdt, N, N, 1, N, 1, None, 8, D, outs[i % 2].data_ptr(), D, stream))
            for i in range(4): fq(i)
            res[f"{str(dt).split('.')[-1]}_{name}"] = {"quantize_us": round(B.time_kernel(fq, 16), 1), "dequantize_us": round(B.time_kernel(fd, 16), 1), "fake_quantize_us": round(B.time_kernel(ff, 16), 1),
                                                      "qd_alg_MB": round(N * N * (es + 1) / 1e6, 1), "fq_alg_MB": round(N * N * 2 * es / 1e6, 1)}
        del ws, qs, outs
        torch.cuda.empty_cache()
    print(json.dumps(res))
elif what == "gidx":
    # W4A16 g128 with activation ordering (weight_g_idx): compress / decompress through the C ABI at 8192^2 bf16
    from compressed_tensors_amd import _lib, codec
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    N = 8192; nsets = 6
    g = torch.Generator(device=dev).manual_seed(5)
    D = _lib.BF16
    perm = torch.randperm(N, device=dev, generator=g)
    g_idx = torch.empty(N, dtype=torch.int32, device=dev); g_idx[perm] = (torch.arange(N, device=dev) // 128).to(torch.int32)
    sets = []
    for _ in range(nsets):
        w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
        sc = (torch.rand(N, N // 128, device=dev, generator=g) * 0.05 + 0.2).to(torch.bfloat16)
        zp = torch.zeros(N, N // 128, dtype=torch.int8, device=dev)
        sets.append((w, sc, zp, torch.empty(N, N // 8, dtype=torch.int32, device=dev), torch.empty(N, N, dtype=torch.bfloat16, device=dev)))
    res = {}
    for name, cg in (("plain", None), ("g_idx", g_idx)):
        cgp = None if cg is None else cg.data_ptr()
        ca = [(w.data_ptr(), D, sc.data_ptr(), D, None, -1, N, N, 1, 128, N // 128, cgp, 4, D, pk.data_ptr(), stream) for (w, sc, zp, pk, out) in sets]
        da = [(pk.data_ptr(), N, N // 8, N, 4, sc.data_ptr(), D, None, -1, 1, 128, N // 128, cgp, out.data_ptr(), D, stream) for (w, sc, zp, pk, out) in sets]
        def c(i): _lib.check(lib.ct_quant_pack(*ca[i % nsets]))
        def d(i): _lib.check(lib.ct_unpack_dequant(*da[i % nsets]))
        for i in range(nsets): c(i)
        res[name] = {"compress_us": round(B.time_kernel(c, 18), 1), "decompress_us": round(B.time_kernel(d, 18), 1)}
        w, sc, zp, pk, out = sets[0]
        ref = codec.fake_quantize_tensor(w, sc, None, num_bits=4, strategy="group", group_size=128, g_idx=None) if cg is None else None
        if cg is None:
            res[name]["round_trip_equals_fake_quantize"] = bool(torch.equal(out, ref))
    print(json.dumps(res))
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
