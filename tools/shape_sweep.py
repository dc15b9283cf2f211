"""W4A16 compress / decompress over real model shapes, group sizes and schemes (HBM-cold): looking for shapes that fall off the lean paths"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compressed_tensors_amd import _lib, codec
lib = _lib.load()
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
BITS = int(os.environ.get("BITS", "4"))
dtype = torch.bfloat16; dt = _lib.DT[dtype]
shapes = [(8192, 8192), (28672, 8192), (8192, 28672), (1024, 8192), (14336, 4096), (4096, 14336), (1024, 4096), (3584, 3584), (18944, 3584), (3584, 18944), (512, 3584),
          (7168, 2048), (2048, 7168), (1536, 7168), (7168, 18432), (5120, 5120), (13824, 5120), (5120, 13824), (11008, 4096), (4096, 11008)]
cfgs = [("g128 sym", 128, True, False), ("g128 asym", 128, False, False)]
if os.environ.get("MORE"):
    cfgs += [("g64 sym", 64, True, False), ("g32 sym", 32, True, False), ("channel sym", None, True, False), ("g128 actorder", 128, True, True)]
for name, group, sym, actorder in cfgs:
    for (r, c) in shapes:
        gs = c if group is None else group
        if c % gs: continue
        nsets = min(600, max(3, -(-(2 * 256 * 2 ** 20) // (r * c * BITS // 8))))  # the smallest read stream (the packed words) >= 2 x the Infinity Cache
        g = torch.Generator(device=dev).manual_seed(31)
        cg = g_idx = None
        if actorder:
            g_idx = (torch.randperm(c, device=dev, generator=g) // gs).to(torch.int32)
            cg = codec.QuantLayout((r, c), torch.empty(r, c // gs, dtype=dtype, device=dev), "group", gs, None, g_idx).col_group
            order = torch.argsort(g_idx)
        sets = []
        for _ in range(nsets):
            w = torch.randn(r, c, dtype=torch.float32, device=dev, generator=g).to(dtype)
            if group is None:
                scale, zp = codec.minmax_qparams(w, num_bits=BITS, group_size=None, symmetric=sym) if hasattr(codec, "minmax_qparams") else (None, None)
            else:
                scale, zp = codec.minmax_qparams(w[:, order].contiguous() if actorder else w, num_bits=BITS, group_size=gs, symmetric=sym)
            sets.append((w, scale, zp, torch.empty(r, c * BITS // 32, dtype=torch.int32, device=dev), torch.empty(r, c, dtype=dtype, device=dev)))
        cgp = None if cg is None else cg.data_ptr()
        sc_cols = sets[0][1].shape[1]
        ca = [(w.data_ptr(), dt, sc.data_ptr(), dt, zp.data_ptr(), _lib.I8, r, c, 1, gs, sc_cols, cgp, BITS, dt, pk.data_ptr(), stream) for (w, sc, zp, pk, out) in sets]
        da = [(pk.data_ptr(), r, pk.shape[1], c, BITS, sc.data_ptr(), dt, None if sym else zp.data_ptr(), -1 if sym else _lib.I8, 1, gs, sc_cols, cgp, out.data_ptr(), dt, stream) for (w, sc, zp, pk, out) in sets]
        for i in range(nsets): _lib.check(lib.ct_quant_pack(*ca[i]))
        for i in range(nsets): _lib.check(lib.ct_unpack_dequant(*da[i]))
        torch.cuda.synchronize()
        w, sc, zp, pk, out = sets[0]
        kw = dict(num_bits=BITS, strategy="group" if group else "channel", group_size=group, g_idx=g_idx)
        ok = torch.equal(out, codec.fake_quantize_tensor(w, sc, zp, **kw))
        alg = 2 * r * c + 2 * r * sc_cols + r * c * BITS // 8 + (0 if sym else r * sc_cols) + (4 * c if actorder else 0)
        res = {}
        n = max(2 * nsets, 60 if r * c > 3e7 else 200)
        for nm, fn, args in (("c", lib.ct_quant_pack, ca), ("d", lib.ct_unpack_dequant, da)):
            ts = []
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(n): fn(*args[i % nsets])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1000 / n)
            res[nm] = sorted(ts)[1]
        flag = "  <<<<" if min(alg / res['c'], alg / res['d']) / 8e6 < 0.55 and alg > 6e7 else ""
        print(f"W{BITS} {name:14s} {r:6d}x{c:<6d} {alg/1e6:7.1f} MB: compress {res['c']:7.1f} us ({alg / res['c'] / 8e6:.3f})  decompress {res['d']:7.1f} us ({alg / res['d'] / 8e6:.3f})  ok={ok}{flag}", flush=True)
        del sets, ca, da
        torch.cuda.empty_cache()
