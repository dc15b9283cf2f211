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
elif what == "bm1":
    # one-pass vs two-kernel sparse compress: correctness on several shapes / densities, then timing at 8192^2
    from compressed_tensors_amd import codec
    import time
    res = {"mode": os.environ.get("CT_BITMASK_ONEPASS")}
    g = torch.Generator(device=dev).manual_seed(5)
    ok = True
    for (r, c, dens) in ((8192, 8192, 0.5), (4096, 4096, 0.5), (1024, 2048, 0.1), (3000, 1000, 0.9), (513, 8200, 0.5), (8192, 8192, 0.0), (2048, 4096, 1.0)):
        w = torch.randn(r, c, dtype=torch.bfloat16, device=dev, generator=g)
        w = w.masked_fill(torch.rand(r, c, device=dev, generator=g) >= dens, 0)
        v, bm, ro = codec.bitmask_compress(w)
        v2, bm2, ro2 = codec.bitmask_compress(w, two_pass=True)
        same = torch.equal(v.view(torch.int16), v2.view(torch.int16)) and torch.equal(bm, bm2) and torch.equal(ro, ro2)
        ok = ok and same
        if not same:
            res.setdefault("bad", []).append([r, c, dens, int(v.numel()), int(v2.numel())])
    res["equal_to_two_pass"] = ok
    r = B.bitmask_leg(dev)
    res.update(compress_us=r["compress_us"], ok=r["round_trip_bit_exact"])
    print(json.dumps(res))
elif what == "bm2":
    r = B.bitmask_leg(dev)
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("CT_BITMASK")}, "compress_us": r["compress_us"], "ok": r["round_trip_bit_exact"]}))
elif what == "bm3":
    # hand-off statistics of the one-pass kernel (CT_BITMASK_OP_NOWAIT=2)
    from compressed_tensors_amd import _lib
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    N = B.N
    g = torch.Generator(device=dev).manual_seed(7)
    w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
    w = w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0)
    ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
    ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
    vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev); bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
    for _ in range(3):
        lib.ct_bitmask_compress(w.data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), ws[-1:].data_ptr(), ws.data_ptr(), ws_bytes, stream)
        torch.cuda.synchronize()
        tiles = N * N // 8 // 1024
        st = ws[tiles + 256: tiles + 259].tolist()
        print(json.dumps({"tiles": tiles, "polls_per_tile": st[0] / tiles, "avg_wait_us": st[1] / tiles / 100.0, "max_wait_us": st[2] / 100.0, "total": int(ws[-1].item())}))
elif what == "bm4":
    # per-tile time stamps of the one-pass kernel (CT_BITMASK_OP_NOWAIT=3)
    from compressed_tensors_amd import _lib
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    N = B.N
    g = torch.Generator(device=dev).manual_seed(7)
    w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
    w = w.masked_fill(torch.rand(N, N, device=dev, generator=g) < 0.5, 0)
    ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
    ws = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
    vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev); bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
    tiles = N * N // 8 // 1024
    for rep in range(3):
        lib.ct_bitmask_compress(w.data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), ws[-1:].data_ptr(), ws.data_ptr(), ws_bytes, stream)
        torch.cuda.synchronize()
    st = ws[tiles + 260: tiles + 260 + 4 * tiles].reshape(tiles, 4).cpu().double()
    t0 = st[:, 0].min()
    st = (st - t0) / 100.0
    import numpy as np
    a = st.numpy()
    print("kernel span us", a[:, 3].max())
    for t in list(range(0, 64, 9)) + list(range(64, tiles, 509)):
        print(t, "start %.1f loaded %.1f resolved %.1f done %.1f" % tuple(a[t]))
    print("median load %.1f wait %.1f scatter %.1f" % (np.median(a[:, 1] - a[:, 0]), np.median(a[:, 2] - a[:, 1]), np.median(a[:, 3] - a[:, 2])))
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
