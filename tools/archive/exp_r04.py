"""round-4 experiments (dev helper, run on the GPU box): python tools/exp_r04.py bmx|bmstamps|marlin|host   (knobs from the environment)

bmx       sparse-bitmask compress at 8192^2 bf16 50 %: HBM-cold time of ct_bitmask_compress (6 rotating inputs, 5 x 60 launches) and a
          parity check against the count / scan / scatter form, under whatever CT_BM_X / CT_BITMASK_RESIDENT says
bmstamps  CT_BITMASK_RESIDENT=3: per-workgroup time stamps (start / published / resolved / done) of the first 1024 workgroups,
          summarised per residency round and per decile of the block index
marlin    marlin-24 fused kernel at 8192^2 (C ABI), parity against the codec's own unfused chain
host      host cost of the plug-in calls that wait on the device (mailbox vs the torch forms they replaced)
"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
what = sys.argv[1]
from compressed_tensors_amd import _lib, codec

if any(k.startswith("CT_BITMASK_RESIDENT") for k in os.environ):
    _lib.LIB_PATH = _lib.DIAG_LIB_PATH  # the knobs exist in the diagnostics build only (-DCT_DIAG)

lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
N = 8192


def sparse_inputs(n, nsets, seed=11):
    g = torch.Generator(device=dev).manual_seed(seed)
    out = []
    for _ in range(nsets):
        w = torch.randn(n, n, dtype=torch.bfloat16, device=dev, generator=g)
        out.append(w.masked_fill(torch.rand(n, n, device=dev, generator=g) < 0.5, 0))
    return out


if what in ("bmx", "bmstamps"):
    res = {"CT_BM_X": os.environ.get("CT_BM_X"), "CT_BITMASK_RESIDENT": os.environ.get("CT_BITMASK_RESIDENT")}
    ws_ = sparse_inputs(N, 6)
    ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
    wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
    vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev)
    bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev)
    ro = torch.empty(N, dtype=torch.int64, device=dev)
    f = lambda i: lib.ct_bitmask_compress(ws_[i % 6].data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(),
                                          wk.data_ptr(), ws_bytes, stream)
    sp = {}
    res["us"] = round(B.time_kernel(f, 60, spread=sp), 2)
    res.update(sp)
    # parity of the last launch (input (60 - 1) % 6 = 5) against the two-pass form
    torch.cuda.synchronize()
    v2, bm2, ro2 = codec.bitmask_compress(ws_[5], two_pass=True)
    nnz = int(wk[-1].item())
    res["ok"] = bool(nnz == v2.numel() and torch.equal(vals[:nnz].view(torch.int16), v2.view(torch.int16)) and torch.equal(bm, bm2) and torch.equal(ro, ro2))
    if what == "bmstamps":
        import numpy as np

        base = 8192 + 4  # kResMaxWGs + control words
        wk[base: base + 4 * 1024] = 0
        f(0)
        torch.cuda.synchronize()
        st = wk[base: base + 4 * 1024].reshape(1024, 4).cpu().double().numpy()
        st = (st - st[:, 0].min()) / 100.0  # us
        names = ("start", "published", "resolved", "done")
        res["stamps_us"] = {}
        for lo in range(0, 1024, 128):
            sl = st[lo: lo + 128]
            res["stamps_us"][f"wg{lo}-{lo + 127}"] = {n: [round(float(np.median(sl[:, k])), 1), round(float(sl[:, k].max()), 1)] for k, n in enumerate(names)}
        res["stamps_us"]["all"] = {"done_max": round(float(st[:, 3].max()), 1), "wait_med_round1": round(float(np.median(st[:512, 2] - st[:512, 1])), 1),
                                   "wait_med_round2": round(float(np.median(st[512:, 2] - st[512:, 1])), 1),
                                   "store_med_round1": round(float(np.median(st[:512, 3] - st[:512, 2])), 1), "store_med_round2": round(float(np.median(st[512:, 3] - st[512:, 2])), 1),
                                   "load_med_round1": round(float(np.median(st[:512, 1] - st[:512, 0])), 1), "load_med_round2": round(float(np.median(st[512:, 1] - st[512:, 0])), 1)}
    print(json.dumps(res))
elif what == "marlin":
    r = B.marlin24_leg(dev)
    print(json.dumps({k: r[k] for k in ("kernels_us", "kernels_frac_hbm", "compress_us_default", "compress_us_deferred_check", "host_issue_us_per_call", "bit_exact_vs_oracle")}))
elif what == "host":
    import compressed_tensors_amd as cta

    out = {}

    def per_call(fn, n=2000):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / n * 1e6, 2)

    s = _lib.stream_of_device(dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    small = torch.zeros(64, dtype=torch.float32, device=dev)
    mb = _lib.mailbox(0)
    out["tiny kernel + .item() (torch D2H + sync)"] = per_call(lambda: (small.add_(1), int(flag.item())))
    out["tiny kernel + ct_stream_wait (spin on hipStreamQuery)"] = per_call(lambda: (small.add_(1), _lib.stream_wait(s)))
    out["tiny kernel + torch.cuda.synchronize()"] = per_call(lambda: (small.add_(1), torch.cuda.synchronize()))
    w = sparse_inputs(256, 1)[0]
    out["codec.bitmask_compress 256x256 (mailbox nnz)"] = per_call(lambda: codec.bitmask_compress(w))
    out["codec.bitmask_compress 256x256 two_pass (.item())"] = per_call(lambda: codec.bitmask_compress(w, two_pass=True))
    print(json.dumps(out, indent=1))
if what == "hostmodel":
    # where the host time of ModelCompressor.compress_model / decompress_model goes on a 154-module TinyLlama-shaped tree (C++ host loop)
    import compressed_tensors_amd as cta
    from compressed_tensors_amd.compressors import base as cbase
    from compressed_tensors_amd.compressors.pack_quantized import base as pq
    from compressed_tensors_amd.quantization.quant_args import QuantizationStatus

    hp = pq._hostpath()
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    g = torch.Generator(device=dev).manual_seed(1)
    mods, keep = [(f"model.layers.{l}.{n}", r, c) for l in range(22) for (n, r, c) in B.TINYLLAMA_LAYER], []
    for _, r, c in mods:
        w = torch.randn(r, c, dtype=torch.bfloat16, device=dev, generator=g)
        s_, z = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
        keep.append((w, s_, z, None, None))
    model = B.tinyllama_module_tree(mods, keep, scheme)
    mc = cta.ModelCompressor()
    for _ in range(3):
        mc.compress_model(model); mc.decompress_model(model)
    torch.cuda.synchronize()

    def med(fn, n=9):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e6)
        return round(sorted(ts)[n // 2], 1), r

    out = {}
    out["walk (C++)"], ms = med(lambda: hp.quantized_modules(model))
    out["walk (named_modules + is_module_quantized)"], _ = med(lambda: mc._named_quantized_modules(model))
    out["_by_format"], groups = med(lambda: cbase._by_format(ms, None))
    infos = [128] * len(ms)
    t_plan, (planned, rest) = med(lambda: hp.w4_plan_compress(ms, infos), n=1)
    out["w4_plan_compress (one call: allocates 154 outputs)"] = t_plan
    (key, (words, n, jobs, _zw, _zn)), = planned.items()
    out["launch_w4_words compress (plan + pinned upload + launch)"], _ = med(lambda: codec.launch_w4_words(words, n, "compress", torch.bfloat16, dev), n=5)
    t0 = time.perf_counter(); hp.w4_finish_compress(jobs, QuantizationStatus.COMPRESSED); out["w4_finish_compress"] = round((time.perf_counter() - t0) * 1e6, 1)
    torch.cuda.synchronize()
    t_plan, (planned, rest) = med(lambda: hp.w4_plan_decompress(ms, [1] * len(ms)), n=1)
    out["w4_plan_decompress"] = t_plan
    (key, (words, n, jobs, _zw, _zn)), = planned.items()
    out["launch_w4_words decompress"], _ = med(lambda: codec.launch_w4_words(words, n, "decompress", torch.bfloat16, dev), n=5)
    t0 = time.perf_counter(); hp.w4_finish_decompress(jobs, QuantizationStatus.DECOMPRESSED); out["w4_finish_decompress"] = round((time.perf_counter() - t0) * 1e6, 1)
    torch.cuda.synchronize()
    out["compress_model (host, until it returns)"], _ = med(lambda: mc.compress_model(model), n=1)
    out["decompress_model (host, until it returns)"], _ = med(lambda: mc.decompress_model(model), n=1)
    print(json.dumps(out, indent=1))
if what == "hostab":
    # native (csrc/host/ct_hostpath.cpp) vs Python host side of the two waiting plug-in calls, and ModelCompressor on the 154-module tree,
    # on one lease: A / B / A / B so that drift between leases does not decide.  (Run Q also compared a streamed head of 16 / 24 / 32 / 48
    # modules in ModelCompressor against one call: 1.151 / 1.142 / 1.144 / 1.128 vs 1.104 ms — removed, profiles/r04_host_native_ab.jsonl)
    import compressed_tensors_amd as cta
    from compressed_tensors_amd.compressors.sparse.sparse_bitmask import BitmaskTensor

    hp = _lib.hostpath()
    ws_ = sparse_inputs(N, 6)
    g = torch.Generator(device=dev).manual_seed(4)
    w24 = []
    for _ in range(6):
        w = torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g)
        w24.append(w.masked_fill_(~codec.sparse24_mask(w), 0))
    sc, zp = codec.minmax_qparams(w24[0], num_bits=4, group_size=128, symmetric=True)
    scheme24 = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, strategy="group", group_size=128, symmetric=True))
    keep = {}

    def bm(i):
        keep["bt"] = BitmaskTensor.from_dense(ws_[i % 6])

    def m24(i):
        keep["m"] = cta.Marlin24Compressor.compress({"weight": w24[i % 6], "weight_scale": sc, "weight_zero_point": zp}, scheme24)

    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    mods, kp = [(f"model.layers.{l}.{n}", r, c) for l in range(22) for (n, r, c) in B.TINYLLAMA_LAYER], []
    for _, r, c in mods:
        w = torch.randn(r, c, dtype=torch.bfloat16, device=dev, generator=g)
        s_, z = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
        kp.append((w, s_, z, None, None))
    model = B.tinyllama_module_tree(mods, kp, scheme)
    mc = cta.ModelCompressor()

    def model_ms(n=9):
        ts = []
        for k in range(n + 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mc.compress_model(model); torch.cuda.synchronize(); t1 = time.perf_counter()
            mc.decompress_model(model); torch.cuda.synchronize(); t2 = time.perf_counter()
            if k >= 2:
                ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
        med = lambda j: round(sorted(t[j] for t in ts)[len(ts) // 2], 4)
        return {"both": med(0), "compress": med(1), "decompress": med(2)}

    out = []
    for rnd in range(2):
        for label, h, mode in (("native", hp, 1), ("native, waits through ct_mailbox_wait_i64 / ct_stream_wait only", hp, 0), ("python", None, 1)):
            _lib._HOSTPATH[0] = h
            hp.set_wait_mode(mode)
            out.append({"host": label, "from_dense_us": [round(v, 2) for v in B.time_calls(bm)], "marlin_default_us": [round(v, 2) for v in B.time_calls(m24)]})
        hp.set_wait_mode(1)
        _lib._HOSTPATH[0] = hp
        out.append({"model_ms": model_ms()})
    for o in out:
        print(json.dumps(o))
if what == "asymab":
    # ModelCompressor on the 154-module tree with the ASYMMETRIC int4 scheme: C++ host loop against the Python loop, A / B / A / B on one lease
    import compressed_tensors_amd as cta

    hp = _lib.hostpath()
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=False, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    g = torch.Generator(device=dev).manual_seed(4)
    mods, kp = [(f"model.layers.{l}.{n}", r, c) for l in range(22) for (n, r, c) in B.TINYLLAMA_LAYER], []
    for _, r, c in mods:
        w = torch.randn(r, c, dtype=torch.bfloat16, device=dev, generator=g)
        s_, z = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=False)
        kp.append((w, s_, z, None, None))
    model = B.tinyllama_module_tree(mods, kp, scheme)
    mc = cta.ModelCompressor()

    def model_ms(n=7):
        ts = []
        for k in range(n + 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mc.compress_model(model); torch.cuda.synchronize(); t1 = time.perf_counter()
            mc.decompress_model(model); torch.cuda.synchronize(); t2 = time.perf_counter()
            if k >= 2:
                ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
        med = lambda j: round(sorted(t[j] for t in ts)[len(ts) // 2], 4)
        return {"both": med(0), "compress": med(1), "decompress": med(2)}

    for rnd in range(2):
        for label, h in (("C++ host loop", hp), ("Python loop", None)):
            _lib._HOSTPATH[0] = h
            print(json.dumps({"host": label, "asymmetric_model_ms": model_ms()}))
    _lib._HOSTPATH[0] = hp
