"""real-shape sweep of the observer kernels, the one-pass round-to-nearest compressions, the unfused pack / unpack and the mixed-dtype
W4 calls (bf16 weight, float32 scale) — HBM-cold, through the Python codec entries; looking for shapes that fall off the lean kernels"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from compressed_tensors_amd import codec
dev = torch.device("cuda:0")
shapes = [(8192, 8192), (28672, 8192), (8192, 28672), (14336, 4096), (4096, 14336), (18944, 3584), (3584, 18944), (7168, 18432), (18432, 7168),
          (5120, 5120), (13824, 5120), (5120, 13824), (11008, 4096), (4096, 11008), (24576, 1536), (4096, 4096)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
which = os.environ.get("WHICH", "obs,rtn,pack,mixed").split(",")


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


def line(name, r, c, alg, us, ok=""):
    f = alg / us / 8e6
    flag = "  <<<<" if f < 0.55 and alg > 6e7 else ""
    print(f"{name:30s} {r:6d}x{c:<6d} {alg/1e6:7.1f} MB: {us:7.1f} us ({f:.3f}) {ok}{flag}", flush=True)


for (r, c) in shapes:
    g = torch.Generator(device=dev).manual_seed(11)
    nsets = min(600, max(3, -(-(2 * 256 * 2 ** 20) // (r * c * 2))))  # the weights themselves are the read stream
    n = max(2 * nsets, 90 if r * c > 3e7 else 200)
    ws = [torch.randn(r, c, device=dev, generator=g, dtype=torch.bfloat16) for _ in range(nsets)]
    W = 2 * r * c
    if "obs" in which:
        for name, fn, sbytes in (
            ("observer int4 g128 sym", lambda i: codec.minmax_qparams(ws[i], num_bits=4, group_size=128, symmetric=True), 3 * r * c // 128),
            ("observer int4 g128 asym", lambda i: codec.minmax_qparams(ws[i], num_bits=4, group_size=128, symmetric=False), 3 * r * c // 128),
            ("observer int8 channel", lambda i: codec.minmax_qparams(ws[i], num_bits=8, group_size=None, symmetric=True), 3 * r),
            ("observer fp8 channel", lambda i: codec.minmax_qparams_float(ws[i], kind="fp8"), 2 * r),
            ("observer mxfp4", lambda i: codec.minmax_qparams_float(ws[i], kind="mxfp4", group_size=32), 2 * r * c // 32),
            ("observer nvfp4", lambda i: codec.minmax_qparams_float(ws[i], kind="nvfp4", group_size=16, global_scale=torch.ones(1, device=dev)), 4 * r * c // 16),
        ):
            if "g128" in name and c % 128: continue
            line(name, r, c, W + sbytes, timeit(fn, nsets, n))
    if "rtn" in which:
        cases = [("rtn W4 g128 sym", lambda i: codec.rtn_quantize_and_pack(ws[i], group_size=128, symmetric=True), W + r * c // 2 + 3 * r * c // 128),
                 ("rtn W4 g128 asym", lambda i: codec.rtn_quantize_and_pack(ws[i], group_size=128, symmetric=False), W + r * c // 2 + 3 * r * c // 128),
                 ("rtn mxfp4", lambda i: codec.rtn_mxfp4_quantize_and_pack(ws[i]), W + r * c // 2 + r * c // 32),
                 ("rtn nvfp4 (given global)", lambda i: codec.rtn_nvfp4_quantize_and_pack(ws[i], torch.full((1,), 448.0, device=dev)), W + r * c // 2 + r * c // 16)]
        if c <= 16384:
            cases += [("rtn int8 channel", lambda i: codec.rtn_quantize_channel8(ws[i], qtype="int", symmetric=True), W + r * c + 3 * r),
                      ("rtn fp8 channel", lambda i: codec.rtn_quantize_channel8(ws[i], qtype="float", symmetric=True), W + r * c + 2 * r)]
        for name, fn, alg in cases:
            if "g128" in name and c % 128: continue
            line(name, r, c, alg, timeit(fn, nsets, n))
    if "pack" in which:
        # the unfused primitives: int8 codes <-> int32 words
        nq = min(600, max(3, -(-(2 * 256 * 2 ** 20) // (r * c // 2))))
        nq = min(nq, max(3, (24 << 30) // (r * c * 2)))
        qs = [torch.randint(-8, 8, (r, c), device=dev, generator=g, dtype=torch.int8) for _ in range(nq)]
        for bits in (4, 8):
            qb = qs if bits == 4 else [q * 16 for q in qs]
            pk = [codec.pack_to_int32(q, bits) for q in qb]
            alg = r * c + r * c * bits // 8
            up = timeit(lambda i: codec.pack_to_int32(qb[i], bits), nq, n)
            uu = timeit(lambda i: codec.unpack_from_int32(pk[i], bits, (r, c)), nq, n)
            ok = torch.equal(codec.unpack_from_int32(pk[0], bits, (r, c)), qb[0])
            line(f"pack_to_int32 b={bits}", r, c, alg, up)
            line(f"unpack_from_int32 b={bits}", r, c, alg, uu, f"ok={ok}")
            del pk
        del qs
    if "mixed" in which and c % 128 == 0:
        # bf16 weight with float32 scales (the quotient and the result are float32, torch's promotion)
        ss = [codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)[0].float() for w in ws]
        kw = dict(num_bits=4, strategy="group", group_size=128)
        pk = [codec.quantize_and_pack(w, s, None, **kw) for w, s in zip(ws, ss)]
        uc = timeit(lambda i: codec.quantize_and_pack(ws[i], ss[i], None, **kw), nsets, n)
        ud = timeit(lambda i: codec.unpack_and_dequantize(pk[i], (r, c), ss[i], None, **kw), nsets, n)
        line("W4 g128 bf16 x, f32 scale: c", r, c, W + r * c // 2 + 4 * r * c // 128, uc)
        line("W4 g128 f32 scale: d (f32 out)", r, c, 4 * r * c + r * c // 2 + 4 * r * c // 128, ud)
        del ss, pk
    del ws
    torch.cuda.empty_cache()
