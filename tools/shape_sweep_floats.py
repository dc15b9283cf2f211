"""real-shape sweep of the strategies / formats the other two sweeps leave out (HBM-cold, through the Python codec entries, whose launches queue behind
one another): FP8 channel / block 128x128 / group 128 / per-tensor, int8 per-tensor / block / asymmetric group, NVFP4, MXFP4, and the W4 path on fp16 and
float32 weights — looking for shapes and layouts that fall off the lean kernels"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compressed_tensors_amd import codec
dev = torch.device("cuda:0")
F8 = torch.float8_e4m3fn
shapes = [(8192, 8192), (28672, 8192), (8192, 28672), (14336, 4096), (4096, 14336), (3584, 3584), (18944, 3584), (3584, 18944), (7168, 2048), (2048, 7168),
          (7168, 18432), (18432, 7168), (5120, 5120), (13824, 5120), (5120, 13824), (11008, 4096), (4096, 11008), (1536, 7168), (24576, 1536)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
which = os.environ.get("WHICH", "fp8,int8,fp4,w4dt").split(",")
WDT = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32}[os.environ.get("DTYPE", "bf16")]  # the weights' (and scales') dtype of the fp8 / int8 / fp4 legs


def timeit(fn, nsets, n):
    for i in range(nsets + 3): fn(i % nsets)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i % nsets)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / n)
    return sorted(ts)[1]


def line(name, r, c, alg_c, uc, alg_d, ud, ok):
    fc, fd = alg_c / uc / 8e6, alg_d / ud / 8e6
    flag = "  <<<<" if min(fc, fd) < 0.55 and alg_c > 6e7 else ""
    print(f"{name:22s} {r:6d}x{c:<6d} {alg_c/1e6:7.1f} MB: compress {uc:7.1f} us ({fc:.3f})  decompress {ud:7.1f} us ({fd:.3f})  ok={ok}{flag}", flush=True)


def scale_like(w, strategy, gs, block, qmax, sdtype):
    r, c = w.shape
    a = w.float().abs()
    if strategy == "tensor":
        s = a.amax().reshape(1)
    elif strategy == "channel":
        s = a.amax(dim=1, keepdim=True)
    elif strategy == "group":
        s = a.reshape(r, c // gs, gs).amax(dim=2)
    else:
        bh, bw = block
        s = a.reshape(r // bh, bh, c // bw, bw).amax(dim=(1, 3))
    return (s / qmax).clamp_min(1e-6).to(sdtype).contiguous()


for (r, c) in shapes:
    g = torch.Generator(device=dev).manual_seed(7)
    # the smallest read stream of any leg here is r x c / 2 bytes (the FP4 codes)
    nsets8 = min(600, max(3, -(-(2 * 256 * 2 ** 20) // (r * c))))
    n8 = max(2 * nsets8, 90 if r * c > 3e7 else 200)
    ws = [torch.randn(r, c, device=dev, generator=g, dtype=torch.float32).to(WDT) for _ in range(nsets8)]
    if "fp8" in which or "int8" in which:
        cases = []
        if "fp8" in which:
            cases += [("fp8 channel", "float", "channel", None, None, True), ("fp8 block128x128", "float", "block", None, [128, 128], True),
                      ("fp8 group128", "float", "group", 128, None, True), ("fp8 tensor", "float", "tensor", None, None, True),
                      ("fp8 group32 (MXFP8's layout)", "float", "group", 32, None, True)]
        if "int8" in which:
            cases += [("int8 tensor", "int", "tensor", None, None, True), ("int8 block128x128", "int", "block", None, [128, 128], True),
                      ("int8 group128 asym", "int", "group", 128, None, False), ("int8 channel asym", "int", "channel", None, None, False)]
        for name, qtype, strategy, gs, block, sym in cases:
            if strategy == "block" and (r % 128 or c % 128): continue
            if strategy == "group" and c % gs: continue
            qmax = 448.0 if qtype == "float" else 127.0
            ss = [scale_like(w, strategy, gs, block, qmax, WDT) for w in ws]
            zs = [None if sym else torch.randint(-20, 20, s.shape, device=dev, generator=g, dtype=torch.int8) for s in ss]
            odt = F8 if qtype == "float" else torch.int8
            kw = dict(num_bits=8, strategy=strategy, group_size=gs, block_structure=block, qtype=qtype)
            qs = [codec.quantize_tensor(w, s, z, dtype=odt, **kw) for w, s, z in zip(ws, ss, zs)]
            uc = timeit(lambda i: codec.quantize_tensor(ws[i], ss[i], zs[i], dtype=odt, **kw), nsets8, n8)
            dkw = dict(strategy=strategy, group_size=gs, block_structure=block)
            ud = timeit(lambda i: codec.dequantize_tensor(qs[i], ss[i], zs[i], **dkw), nsets8, n8)
            back = codec.dequantize_tensor(qs[0], ss[0], zs[0], **dkw)
            ok = torch.equal(back, codec.fake_quantize_tensor(ws[0], ss[0], zs[0], **kw))
            es = ws[0].element_size()
            alg = (es + 1) * r * c + ss[0].numel() * (es + (0 if sym else 1))
            line(name, r, c, alg, uc, alg, ud, ok)
            del qs, ss, zs
    if "fp4" in which and c % 32 == 0:
        for fmt, group in (("nvfp4", 16), ("mxfp4", 32)):
            if fmt == "nvfp4":
                ss = [(torch.rand((r, c // group), device=dev, generator=g) * 2 + 0.1).to(F8).float() for _ in range(nsets8)]
                gsc = torch.tensor([3.7], device=dev)
                kind = "f8e4m3"
            else:
                ss = [torch.full((r, c // group), 0.5, device=dev, dtype=WDT) for _ in range(nsets8)]
                gsc, kind = None, "e8m0"
            ps = [codec.fp4_quantize_and_pack(x, s, gsc, group_size=group) for x, s in zip(ws, ss)]
            cs = [s.to(F8) if fmt == "nvfp4" else torch.full(s.shape, 126, dtype=torch.uint8, device=dev) for s in ss]
            nsets = min(nsets8 * 2, len(ws))  # the packed codes are r x c / 2 bytes: the rotation below is as cold as this leg's sets allow
            uc = timeit(lambda i: codec.fp4_quantize_and_pack(ws[i], ss[i], gsc, group_size=group), nsets8, n8)
            ud = timeit(lambda i: codec.fp4_unpack_and_dequantize(ps[i], cs[i], gsc, group_size=group, scale_kind=kind), nsets8, n8)
            es = ws[0].element_size()
            alg_c = r * c * (es + 0.5) + ss[0].numel() * ss[0].element_size()
            alg_d = r * c * (es + 0.5) + ss[0].numel()
            line(fmt, r, c, alg_c, uc, alg_d, ud, True)
            del ss, ps, cs
    if "w4dt" in which and c % 128 == 0:
        for dtype, tag in ((torch.float16, "W4 g128 fp16"), (torch.float32, "W4 g128 float32")):
            es = 2 if dtype == torch.float16 else 4
            nsets = min(nsets8, max(3, (20 << 30) // (r * c * es * 2)))
            wd = [w.to(dtype) for w in ws[:nsets]]
            sz = [codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True) for w in wd]
            kw = dict(num_bits=4, strategy="group", group_size=128)
            pk = [codec.quantize_and_pack(w, s, None, **kw) for w, (s, z) in zip(wd, sz)]
            uc = timeit(lambda i: codec.quantize_and_pack(wd[i], sz[i][0], None, **kw), nsets, n8)
            ud = timeit(lambda i: codec.unpack_and_dequantize(pk[i], (r, c), sz[i][0], None, **kw), nsets, n8)
            ok = torch.equal(codec.unpack_and_dequantize(pk[0], (r, c), sz[0][0], None, **kw), codec.fake_quantize_tensor(wd[0], sz[0][0], None, **kw))
            alg = r * c * es + r * c // 2 + r * (c // 128) * es
            line(tag, r, c, alg, uc, alg, ud, ok)
            del wd, sz, pk
    del ws
    torch.cuda.empty_cache()
