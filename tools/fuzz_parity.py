"""Randomised differential test of the HIP path against the CPU oracle (developer tool; the fixed cases live in tests/).
    python tools/fuzz_parity.py [seconds] [seed] [max_cases]
Random shapes / dtypes / strategies / special values through the public codec entry points; stops at the first mismatch."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle as O
from compressed_tensors_amd import codec
from test_oracle_golden import eq, eq_f8

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
max_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 62  # the -m gpu test runs a bounded, seeded number of cases
rng = random.Random(seed)
dev = torch.device("cuda:0")
BF16, F16, F32, F8 = torch.bfloat16, torch.float16, torch.float32, torch.float8_e4m3fn
SPECIAL = [0.0, -0.0, 0.5, 1.5, 2.5, -0.5, -2.5, 7.5, -8.5, 127.5, -128.5, 448.0, 464.0, 6.0, 1e-8, -1e-8, 3e4, float("inf"), -float("inf"), float("nan")]


def rand_x(shape, dt, g, allow_nonfinite=True):
    x = (torch.randn(shape, generator=g) * 10 ** rng.uniform(-3, 2)).to(dt)
    n = rng.randint(0, min(x.numel(), len(SPECIAL)))
    if n:
        vals = [v for v in SPECIAL if allow_nonfinite or v == v and abs(v) != float("inf")]
        idx = torch.randint(0, x.numel(), (n,), generator=g)
        x.view(-1)[idx] = torch.tensor([rng.choice(vals) for _ in range(n)], dtype=torch.float32).to(dt)
    return x


def case_quant(g):
    dt = rng.choice([BF16, F16, F32])
    sdt = rng.choice([dt, F32])
    strategy = rng.choice(["tensor", "channel", "group", "group", "block"])
    bits = rng.choice([2, 3, 4, 4, 4, 5, 8, 8])
    qtype = rng.choice(["int", "int", "float"])
    if qtype == "float":
        bits = 8
    rows = rng.choice([1, 3, 8, 33, 64, 200])
    gs = rng.choice([16, 32, 64, 128]) if strategy == "group" else None
    cols = (rng.choice([1, 2, 3, 5, 8, 17]) * (gs or 8)) if strategy == "group" else rng.choice([8, 24, 40, 96, 200, 256, 1000, 1024])
    block = None
    if strategy == "block":
        bh, bw = rng.choice([(4, 32), (8, 16), (16, 64)])
        rows, cols = bh * rng.randint(1, 5), bw * rng.randint(1, 6)
        block = [bh, bw]
    x = rand_x((rows, cols), dt, g)
    sshape = {"tensor": (1,), "channel": (rows, 1), "group": (rows, cols // (gs or 1)) if gs else None, "block": (rows // block[0], cols // block[1]) if block else None}[strategy]
    s = (torch.rand(sshape, generator=g) * 10 ** rng.uniform(-3, 1) + 1e-4).to(sdt)
    sym = rng.random() < 0.5
    if qtype == "float":
        z = rng.choice([None, torch.zeros(sshape, dtype=F8)])
    else:
        z = torch.zeros(sshape, dtype=torch.int8) if sym else torch.randint(-2 ** (bits - 1), 2 ** (bits - 1), sshape, generator=g, dtype=torch.int8)
    kw = dict(num_bits=bits, strategy=strategy, group_size=gs, block_structure=block, qtype=qtype)
    d = lambda t: None if t is None else t.to(dev)
    if strategy == "group" and qtype == "int" and rng.random() < 0.4:
        return case_quant_gidx(x, s, z, kw, dt, sdt, g)
    odt = F8 if qtype == "float" else torch.int8
    q = codec.quantize_tensor(d(x), d(s), d(z), dtype=odt, **kw)
    qr = O.quantize(x, s, z, dtype=odt, **kw)
    assert (eq_f8(q.cpu(), qr) if qtype == "float" else torch.equal(q.cpu(), qr)), ("quantize", kw, dt, sdt, x.shape)
    fq = codec.fake_quantize_tensor(d(x), d(s), d(z), **kw)
    assert eq(fq.cpu(), O.fake_quantize(x, s, z, **kw)), ("fake_quantize", kw, dt, sdt, x.shape)
    dkw = {k: v for k, v in kw.items() if k not in ("num_bits", "qtype")}
    dq = codec.dequantize_tensor(d(qr), d(s), d(z), **dkw)
    assert eq(dq.cpu(), O.dequantize(qr, s, z, **dkw)), ("dequantize", kw, dt, sdt, x.shape)
    if qtype == "int" and strategy != "block":
        packed = codec.quantize_and_pack(d(x), d(s), d(z), **{k: v for k, v in kw.items() if k != "qtype"})
        assert torch.equal(packed.cpu(), O.pack_to_int32(O.quantize(x, s, z, dtype=torch.int8, **kw), bits).contiguous()), ("quant_pack", kw, dt, sdt, x.shape)
        back = codec.unpack_and_dequantize(packed, x.shape, d(s), d(z), num_bits=bits, strategy=strategy, group_size=gs)
        assert eq(back.cpu(), O.dequantize(O.quantize(x, s, z, dtype=torch.int8, **kw), s, z, **dkw)), ("unpack_dequant", kw, dt, sdt, x.shape)


def case_quant_gidx(x, s, z, kw, dt, sdt, g):
    """activation ordering: the same checks with a random column -> group table"""
    cols, gs, bits = x.shape[1], kw["group_size"], kw["num_bits"]
    g_idx = (torch.arange(cols, dtype=torch.int32) // gs)[torch.randperm(cols, generator=g)].contiguous()
    kw = {k: v for k, v in kw.items() if k not in ("qtype", "block_structure")}
    d = lambda t: None if t is None else t.to(dev)
    qr = O.quantize(x, s, z, dtype=torch.int8, g_idx=g_idx, **kw)
    q = codec.quantize_tensor(d(x), d(s), d(z), dtype=torch.int8, g_idx=d(g_idx), **kw)
    assert torch.equal(q.cpu(), qr), ("quantize g_idx", kw, dt, sdt, x.shape)
    assert eq(codec.fake_quantize_tensor(d(x), d(s), d(z), g_idx=d(g_idx), **kw).cpu(), O.fake_quantize(x, s, z, g_idx=g_idx, **kw)), ("fake_quantize g_idx", kw, dt, sdt, x.shape)
    packed = codec.quantize_and_pack(d(x), d(s), d(z), g_idx=d(g_idx), **kw)
    assert torch.equal(packed.cpu(), O.pack_to_int32(qr, bits).contiguous()), ("quant_pack g_idx", kw, dt, sdt, x.shape)
    back = codec.unpack_and_dequantize(packed, x.shape, d(s), d(z), g_idx=d(g_idx), **kw)
    assert eq(back.cpu(), O.dequantize(qr, s, z, strategy="group", group_size=gs, g_idx=g_idx)), ("unpack_dequant g_idx", kw, dt, sdt, x.shape)


def case_pack(g):
    bits = rng.randint(1, 8)
    rows, cols = rng.choice([1, 5, 32, 77]), rng.choice([1, 7, 32, 33, 64, 100, 256, 1024])
    v = torch.randint(-2 ** (bits - 1), 2 ** (bits - 1), (rows, cols), generator=g, dtype=torch.int8)
    p = codec.pack_to_int32(v.to(dev), bits)
    assert torch.equal(p.cpu(), O.pack_to_int32(v, bits).contiguous()), ("pack", bits, rows, cols)
    assert torch.equal(codec.unpack_from_int32(p, bits, v.shape).cpu(), v), ("unpack", bits, rows, cols)


def case_bitmask(g):
    dt = rng.choice([BF16, F16, F32, torch.int8])
    rows, cols = rng.choice([1, 3, 17, 64, 300]), rng.choice([1, 8, 24, 40, 64, 96, 1000, 2080, 4096, 8160, 8192, 8200, 16384 + 32])
    x = torch.randn((rows, cols), generator=g)
    x = x.masked_fill(torch.rand((rows, cols), generator=g) < rng.choice([0.0, 0.1, 0.5, 0.9, 1.0]), 0)
    x = (x * 50).to(dt) if dt == torch.int8 else x.to(dt)
    if dt != torch.int8 and rng.random() < 0.5:  # -0.0 (a zero), NaN / inf / denormals (kept) at random places
        n = rng.randint(1, min(x.numel(), 24))
        idx = torch.randint(0, x.numel(), (n,), generator=g)
        x.view(-1)[idx] = torch.tensor([rng.choice(SPECIAL + [1e-40, -1e-41, 6e-8, -6e-8]) for _ in range(n)], dtype=torch.float32).to(dt)
    values, bitmask, ro = codec.bitmask_compress(x.to(dev), two_pass=rng.random() < 0.3)
    rv, rb, rro = O.bitmask_compress(x)
    n = rv.numel()
    assert torch.equal(values.cpu()[:n].view(torch.uint8), rv.view(torch.uint8)) and torch.equal(bitmask.cpu(), rb) and torch.equal(ro.cpu(), rro), ("bitmask_compress", dt, rows, cols)
    back = codec.bitmask_decompress(values, bitmask, x.shape, ro)
    want = torch.where(x != 0, x, torch.zeros_like(x))  # a dropped -0.0 comes back as +0.0; NaN payloads and infinities come back bit for bit
    assert torch.equal(back.cpu().view(torch.uint8), want.view(torch.uint8).reshape(back.cpu().view(torch.uint8).shape)), ("bitmask_decompress", dt, rows, cols)


def case_fp4(g):
    dt = rng.choice([BF16, F16])
    fmt, group = rng.choice([("nvfp4-pack-quantized", 16), ("mxfp4-pack-quantized", 32)])
    rows, cols = rng.choice([1, 3, 16, 65]), group * rng.choice([1, 2, 3, 5, 8, 64])
    x = rand_x((rows, cols), dt, g, allow_nonfinite=False)
    if fmt.startswith("nvfp4"):
        gs = O.generate_gparam(x)
        s = O.calculate_qparams_float(x, kind="nvfp4", group_size=16, global_scale=gs)
        p, s8, gsd, sd = codec.rtn_nvfp4_quantize_and_pack(x.to(dev), return_scale=True) if cols % 32 == 0 else (None, None, None, None)
    else:
        gs = None
        s = O.calculate_qparams_float(x, kind="mxfp4", group_size=32)
        p, code, sd = codec.rtn_mxfp4_quantize_and_pack(x.to(dev), return_scale=True)
    ref = O.fp4_compress(x, s, gs, fmt=fmt)
    got = codec.fp4_quantize_and_pack(x.to(dev), s.to(dev), None if gs is None else gs.to(dev), group_size=group)
    assert torch.equal(got.cpu(), ref["weight_packed"]), ("fp4 compress", fmt, dt, rows, cols)
    if p is not None:
        assert torch.equal(p.cpu(), ref["weight_packed"]) and eq(sd.cpu(), s), ("fp4 rtn", fmt, dt, rows, cols)
    kind = "f8e4m3" if gs is not None else "e8m0"
    dec = codec.fp4_unpack_and_dequantize(got, ref["weight_scale"].to(dev), None if gs is None else gs.to(dev), group_size=group, scale_kind=kind)
    assert eq(dec.cpu(), O.fp4_decompress(ref, fmt=fmt)["weight"]), ("fp4 decompress", fmt, dt, rows, cols)


def case_rtn(g):
    dt = rng.choice([BF16, F16])
    sym = rng.random() < 0.5
    gs = rng.choice([32, 64, 128, 256, None])
    rows, cols = rng.choice([1, 4, 37]), (gs or 32) * rng.choice([1, 2, 8]) if gs else rng.choice([32, 512, 2048])
    x = rand_x((rows, cols), dt, g, allow_nonfinite=False)
    packed, scale, zp = codec.rtn_quantize_and_pack(x.to(dev), group_size=gs, symmetric=sym)
    s, z = O.calculate_qparams_minmax(x, num_bits=4, group_size=gs, symmetric=sym)
    assert eq(scale.cpu(), s) and torch.equal(zp.cpu(), z), ("rtn qparams", dt, sym, gs, rows, cols)
    q = O.quantize(x, s, z, num_bits=4, strategy="group" if gs else "channel", group_size=gs, dtype=torch.int8)
    assert torch.equal(packed.cpu(), O.pack_to_int32(q, 4).contiguous()), ("rtn pack", dt, sym, gs, rows, cols)



def case_qparams_float(g):
    dt = rng.choice([BF16, F16, F32])
    kind, gs = rng.choice([("fp8", None), ("fp8", 64), ("nvfp4", 16), ("mxfp4", 32), ("mxfp8", 32)])
    rows, cols = rng.choice([1, 5, 40]), (gs or 8) * rng.choice([1, 3, 16, 64])
    x = rand_x((rows, cols), dt, g)
    gsc = O.generate_gparam(torch.nan_to_num(x.float(), nan=0.0, posinf=1.0, neginf=-1.0).to(dt)) if kind == "nvfp4" else None
    got = codec.minmax_qparams_float(x.to(dev), kind=kind, group_size=gs, global_scale=None if gsc is None else gsc.to(dev))
    assert eq(got.cpu(), O.calculate_qparams_float(x, kind=kind, group_size=gs, global_scale=gsc)), ("qparams_float", kind, dt, rows, cols)
    xf = rand_x((rows, cols), dt, g, allow_nonfinite=False)
    assert eq(codec.generate_gparam(xf.to(dev)).cpu(), O.generate_gparam(xf)), ("gparam", dt, rows, cols)


def case_channel8(g):
    dt = rng.choice([BF16, F16])
    qtype, sym = rng.choice([("int", True), ("int", False), ("float", True)])
    rows, cols = rng.choice([1, 6, 50]), 8 * rng.choice([1, 5, 64, 256, 1000, 2048])
    x = rand_x((rows, cols), dt, g, allow_nonfinite=False)
    q, scale, zp = codec.rtn_quantize_channel8(x.to(dev), qtype=qtype, symmetric=sym)
    if qtype == "int":
        s, z = O.calculate_qparams_minmax(x, num_bits=8, group_size=None, symmetric=sym)
        assert eq(scale.cpu(), s) and torch.equal(zp.cpu(), z), ("channel8 qparams", dt, sym, rows, cols)
        assert torch.equal(q.cpu(), O.quantize(x, s, z, num_bits=8, strategy="channel", dtype=torch.int8)), ("channel8 int", dt, sym, rows, cols)
    else:
        s = O.calculate_qparams_float(x, kind="fp8")
        ref = O.quantize(x, s, torch.zeros_like(s, dtype=F8), num_bits=8, strategy="channel", dtype=F8, qtype="float")
        assert eq(scale.cpu(), s) and eq_f8(q.cpu(), ref), ("channel8 fp8", dt, rows, cols)


def case_sparse24(g):
    dt = rng.choice([BF16, F16, torch.int8, F32])
    rows, cols = rng.choice([1, 4, 33, 64]), 8 * rng.choice([1, 2, 9, 64, 130, 260, 512, 1020, 1024, 2056])
    x = torch.randn((rows, cols), generator=g)
    x = (x * 40).to(dt) if dt == torch.int8 else x.to(dt)
    mask = codec.sparse24_mask(x.to(dev))
    assert torch.equal(mask.cpu(), O.sparse24_mask(x)), ("sparse24 mask", dt, rows, cols)
    v, b = codec.sparse24_bitmask_compress(x.to(dev))
    rv, rb = O.sparse24_bitmask_compress(x)
    assert torch.equal(v.cpu().view(torch.uint8), rv.view(torch.uint8)) and torch.equal(b.cpu(), rb), ("sparse24 compress", dt, rows, cols)
    back = codec.sparse24_bitmask_decompress(v, b, x.shape)
    assert torch.equal(back.cpu().view(torch.uint8), O.sparse24_bitmask_decompress(rv, rb, x.shape).view(torch.uint8)), ("sparse24 decompress", dt, rows, cols)


def case_marlin(g):
    import compressed_tensors_amd as cta
    bits, strategy, gs = rng.choice([(4, "group", 128), (4, "channel", None), (8, "channel", None), (4, "group", 64)])
    rows, cols = 64 * rng.choice([1, 2, 3]), 256 * rng.choice([1, 2, 3]) if rng.random() < 0.7 else 64 * rng.choice([1, 3, 5])
    if gs and cols % gs:
        return
    dt = rng.choice([BF16, F16])
    w = (torch.randn((rows, cols), generator=g) * 10 ** rng.uniform(-2, 1)).to(dt)
    w = w * O.sparse24_mask(w).to(w.dtype)
    scale, zp = O.calculate_qparams_minmax(w.to(F16), num_bits=bits, group_size=gs, symmetric=True)
    ref = O.marlin24_compress(w, scale, zp, num_bits=bits, strategy=strategy, group_size=gs)
    args = cta.QuantizationArgs(num_bits=bits, strategy=strategy, group_size=gs, symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    got = cta.Marlin24Compressor.compress({"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}, scheme)
    for k in ref:
        assert eq(got[k].cpu().contiguous(), ref[k].contiguous()), ("marlin24", k, bits, strategy, gs, dt, rows, cols)


def case_batches(g):
    """the one-launch tables: W4A16 (symmetric / asymmetric incl. the batched zero-point packing) and the 8-bit codecs, random module lists"""
    dt = rng.choice([BF16, F16])
    kind = rng.choice(["w4", "w4", "int8", "fp8"])
    n = rng.randint(1, 6)
    ents, dents, refs, zps, carried = [], [], [], [], []
    bits = 4 if kind == "w4" else 8 if kind == "fp8" else rng.choice([8, 8, 6])
    for _ in range(n):
        rows = rng.choice([1, 3, 32, 33, 200])
        if kind == "w4":
            group = rng.choice([32, 64, 128, 256])
            cols = group * rng.choice([1, 2, 5])
            if rng.random() < 0.25:
                group = cols
        else:
            group = rng.choice([16, 64, 128])
            cols = group * rng.choice([1, 3, 8])
            r = rng.random()
            group = cols if r < 0.3 else rows * cols if r < 0.5 else group
        x = rand_x((rows, cols), dt, g, allow_nonfinite=rng.random() < 0.3)
        strategy = "tensor" if group == rows * cols and kind != "w4" else "channel" if group == cols else "group"
        sym = rng.random() < 0.5 or kind == "fp8"
        fin = torch.nan_to_num(x.float(), nan=0.0, posinf=1.0, neginf=-1.0).to(dt)
        if kind == "fp8":
            s = O.calculate_qparams_float(fin.reshape(1, -1) if strategy == "tensor" else fin, kind="fp8", group_size=group if strategy == "group" else None)
            z = None
        else:
            s, z = O.calculate_qparams_minmax(fin.reshape(1, -1) if strategy == "tensor" else fin, num_bits=bits, group_size=group if strategy == "group" else None, symmetric=sym)
            if sym and rng.random() < 0.5:
                z = None
        if strategy == "tensor":
            s = s.reshape(1)
            z = None if z is None else z.reshape(1)
        kw = dict(num_bits=bits, strategy=strategy, group_size=group if strategy == "group" else None)
        xd, sd, zd = x.to(dev), s.to(dev), None if z is None else z.to(dev)
        if kind == "w4":
            q = O.quantize(x, s, z, dtype=torch.int8, **kw)
            refs.append((O.pack_to_int32(q, 4).contiguous(), O.dequantize(q, s, z)))
            packed = torch.empty((rows, cols // 8), dtype=torch.int32, device=dev)
            out = torch.empty((rows, cols), dtype=dt, device=dev)
            carry = z is not None and rng.random() < 0.6  # round 6: the stored zero points ride in the weights' launches (ct_w4_item.zp_packed)
            zpp = torch.full((-(-rows * 4 // 32), z.shape[1]), -1, dtype=torch.int32, device=dev) if carry else None
            ents.append((xd, sd, zd, packed, rows, cols, group, zpp))
            if carry and codec.w4_packed_zp_readable(cols, group):
                back = torch.full_like(zd, 77)
                dents.append((packed, sd, back, out, rows, cols, group, zpp))
                carried.append((zd, zpp, back, O.pack_to_int32(z, 4, packed_dim=0).contiguous()))
            else:
                dents.append((packed, sd, zd, out, rows, cols, group))
                if carry:
                    carried.append((zd, zpp, None, O.pack_to_int32(z, 4, packed_dim=0).contiguous()))
            if z is not None:
                zp_packed = torch.empty((-(-rows * 4 // 32), z.shape[1]), dtype=torch.int32, device=dev)
                zps.append((zd, zp_packed, torch.empty_like(zd), O.pack_to_int32(z, 4, packed_dim=0).contiguous()))
        else:
            qdt = F8 if kind == "fp8" else torch.int8
            q = O.quantize(x, s, z, dtype=qdt, qtype="float" if kind == "fp8" else "int", **kw)
            refs.append((q, O.dequantize(q, s, z)))
            qd = torch.empty((rows, cols), dtype=qdt, device=dev)
            out = torch.empty((rows, cols), dtype=dt, device=dev)
            ents.append((xd, sd, zd, qd, rows, cols, group)); dents.append((qd, sd, zd, out, rows, cols, group))
    codec.W4Batch(ents, "compress", dt, kind=kind, bits=bits).launch()
    codec.W4Batch(dents, "decompress", dt, kind=kind).launch()
    for (xd, sd, zd, code, rows, cols, group, *_), (_, _, _, out, *_), (rq, rd) in zip(ents, dents, refs):
        ok = eq_f8(code.cpu(), rq) if kind == "fp8" else torch.equal(code.cpu(), rq)
        assert ok and eq(out.cpu(), rd), ("batch", kind, dt, bits, rows, cols, group, zd is None)
    for zd, zpp, back, ref in carried:
        assert torch.equal(zpp.cpu(), ref) and (back is None or torch.equal(back, zd)), ("zp carried in the launch", tuple(zd.shape))
    if zps:
        codec.zp4_batch([(a, b) for a, b, _, _ in zps], "pack")
        codec.zp4_batch([(b, c) for _, b, c, _ in zps], "unpack")
        for a, b, c, ref in zps:
            assert torch.equal(b.cpu(), ref) and torch.equal(c, a), ("zp batch", tuple(a.shape))


def case_w4_zp(g):
    """round 6: the single-module asymmetric W4 entries (stored zero points written / read by the weights' own launch) through the class"""
    import compressed_tensors_amd as cta

    dt = rng.choice([BF16, F16])
    group = rng.choice([32, 64, 128, 128, 128, None])
    rows = rng.choice([1, 7, 8, 20, 64, 129])
    cols = (group or 32) * rng.choice([1, 2, 4, 8, 16])
    x = rand_x((rows, cols), dt, g, allow_nonfinite=rng.random() < 0.3)
    fin = torch.nan_to_num(x.float(), nan=0.0, posinf=1.0, neginf=-1.0).to(dt)
    s, z = O.calculate_qparams_minmax(fin, num_bits=4, group_size=group, symmetric=False)
    kw = dict(num_bits=4, strategy="group" if group else "channel", group_size=group)
    ref_c = O.pack_quantized_compress({"weight": x, "weight_scale": s, "weight_zero_point": z}, symmetric=False, **kw)
    ref_d = O.pack_quantized_decompress(ref_c, num_bits=4, strategy=kw["strategy"], symmetric=False)
    args = cta.QuantizationArgs(num_bits=4, group_size=group, symmetric=False, strategy=kw["strategy"])
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    got_c = cta.PackedQuantizationCompressor.compress({"weight": x.to(dev), "weight_scale": s.to(dev), "weight_zero_point": z.to(dev)}, scheme)
    assert torch.equal(got_c["weight_packed"].cpu(), ref_c["weight_packed"]) and torch.equal(got_c["weight_zero_point"].cpu(), ref_c["weight_zero_point"]), ("w4_zp compress", dt, rows, cols, group)
    got_d = cta.PackedQuantizationCompressor.decompress(got_c, scheme)
    assert eq(got_d["weight"].cpu(), ref_d["weight"]) and torch.equal(got_d["weight_zero_point"].cpu(), ref_d["weight_zero_point"]), ("w4_zp decompress", dt, rows, cols, group)


def case_bitmask_many(g):
    """round 6: a list of tensors through ONE table launch (8- / 16- / 32-bit payloads, exact-size results by the batched copy)"""
    xs = []
    for _ in range(rng.randint(1, 7)):
        dt = rng.choice([BF16, F16, F32, torch.int8, BF16])
        rows, cols = rng.choice([1, 3, 17, 64, 300]), rng.choice([8, 16, 24, 48, 64, 1008, 4096, 8192])
        x = torch.randn((rows, cols), generator=g)
        x = x.masked_fill(torch.rand((rows, cols), generator=g) < rng.choice([0.0, 0.1, 0.5, 0.9, 1.0]), 0)
        xs.append((x * 50).to(dt) if dt == torch.int8 else x.to(dt))
    exact = rng.random() < 0.7
    got = codec.bitmask_compress_many([x.to(dev) for x in xs], exact=exact, arena_bytes=rng.choice([1, 1 << 16, 1 << 30]))
    for x, (v, bm, ro) in zip(xs, got):
        rv, rb, rro = O.bitmask_compress(x)
        assert v.numel() == rv.numel() and torch.equal(v.cpu().view(torch.uint8), rv.view(torch.uint8)) and torch.equal(bm.cpu(), rb) and torch.equal(ro.cpu(), rro), ("bitmask many", x.dtype, tuple(x.shape))
        if exact:
            assert v.untyped_storage().nbytes() <= rv.numel() * x.element_size() + (2 << 20)


CASES = [case_quant, case_quant, case_quant, case_pack, case_bitmask, case_bitmask, case_fp4, case_rtn, case_qparams_float, case_channel8, case_sparse24, case_marlin, case_batches, case_batches, case_w4_zp, case_bitmask_many]
t0, n = time.time(), 0
while time.time() - t0 < budget and n < max_cases:
    g = torch.Generator().manual_seed(rng.randint(0, 2 ** 31))
    rng.choice(CASES)(g)
    n += 1
print(f"fuzz: {n} random cases, no mismatch (seed {seed}, {time.time() - t0:.0f} s of {budget:.0f} s)")
