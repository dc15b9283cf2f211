"""Pins the CPU oracle (oracle/ct_oracle.c) against golden vectors produced by the upstream
reference itself (oracle/gen_golden.py) and against the known-answer vectors of the
reference's own tests.  CPU only."""
import math

import pytest
import torch
from _golden import cases

import oracle as O

BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32


def eq(a, b):
    """bitwise tensor equality (NaN == NaN, -0.0 != +0.0)"""
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype.is_floating_point:
        it = {2: torch.int16, 4: torch.int32}[a.element_size()]
        an, bn = a != a, b != b
        if not torch.equal(an, bn):
            return False
        a = torch.where(an, torch.zeros_like(a), a)
        b = torch.where(bn, torch.zeros_like(b), b)
        return torch.equal(a.contiguous().view(it), b.contiguous().view(it))
    return torch.equal(a, b)


# ----------------------------------------------------------------------------- pack / unpack
@pytest.mark.parametrize("case", cases("pack"), ids=lambda c: c["key"])
def test_pack_unpack_golden(golden, case):
    t = golden.case("pack", case["key"])
    bits, pd = case["bits"], case["packed_dim"]
    packed = O.pack_to_int32(t["value"], bits, packed_dim=pd)
    assert eq(packed.contiguous(), t["packed"])
    if not case.get("oob"):
        un = O.unpack_from_int32(t["packed"], bits, torch.Size(case["shape"]), packed_dim=pd)
        assert eq(un.contiguous(), t["value"])


def _old_pack(value, bits):
    """element-aligned historical layout (32//bits codes per word), restated from the
    description in reference tests/test_compressors/test_pack_quant.py:27-39"""
    pf = 32 // bits
    u = (value.to(torch.int32) + (1 << (bits - 1))) & ((1 << bits) - 1)
    rows, cols = u.shape
    padded = math.ceil(cols / pf) * pf
    u = torch.nn.functional.pad(u, (0, padded - cols))
    out = torch.zeros(rows, padded // pf, dtype=torch.int32)
    for i in range(pf):
        out |= u[:, i::pf] << (i * bits)
    return out


@pytest.mark.parametrize("bits", [1, 2, 4, 8])
@pytest.mark.parametrize("k", [33, 64, 100, 1024])
def test_old_format_compat(bits, k):
    """reference tests/test_compressors/test_pack_quant.py:386-416"""
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    v = torch.randint(lo, hi + 1, (64, k), dtype=torch.int8)
    old = _old_pack(v, bits)
    assert torch.equal(O.pack_to_int32(v, bits), old)
    assert torch.equal(O.unpack_from_int32(old, bits, torch.Size((64, k))), v)


def test_pack_errors():
    with pytest.raises(ValueError):
        O.pack_to_int32(torch.zeros(2, 2, dtype=torch.int32), 4)
    with pytest.raises(ValueError):
        O.pack_to_int32(torch.zeros(2, 2, dtype=torch.int8), 9)
    with pytest.raises(ValueError):
        O.unpack_from_int32(torch.zeros(2, 2, dtype=torch.int8), 4, (2, 2))


# ----------------------------------------------------------------------------- quantization
def _kw(case):
    a = case["args"]
    return dict(num_bits=a["num_bits"], strategy=a["strategy"], group_size=a.get("group_size"),
                block_structure=a.get("block_structure"))


@pytest.mark.parametrize("case", cases("quant"), ids=lambda c: c["key"])
def test_quant_golden(golden, case):
    t = golden.case("quant", case["key"])
    kw = _kw(case)
    g_idx = t.get("g_idx")
    q8 = O.quantize(t["x"], t["scale"], t["zp"], dtype=torch.int8, g_idx=g_idx, **kw)
    assert eq(q8, t["q8"])
    fq = O.fake_quantize(t["x"], t["scale"], t["zp"], g_idx=g_idx, **kw)
    assert eq(fq, t["fq"])
    dkw = dict(kw)
    dkw.pop("num_bits")
    dq = O.dequantize(t["q8"], t["scale"], t["zp"], g_idx=g_idx, **dkw)
    assert eq(dq, t["dq"])
    if "qf" in t:
        assert eq(O.quantize(t["x"], t["scale"], t["zp"], g_idx=g_idx, **kw), t["qf"])
        assert eq(O.quantize(t["x"], t["scale"], None, dtype=torch.int8, **kw), t["q8_nozp"])
        if kw["strategy"] != "block":
            assert eq(O.dequantize(t["q8"], t["scale"], t["zp"]), t["dq_inferred"])


# known-answer vectors: reference tests/test_quantization/lifecycle/test_static_lifecycle.py:18-165
_KA = [
    ("tensor", None, [0.0], [23.0],
     [[0.0000, 0.0000, 3.0625, 3.0625, 3.0625, 6.1250], [6.1250, 6.1250, 9.1875, 9.1875, 9.1875, 12.2500],
      [12.2500, 12.2500, 15.3125, 15.3125, 15.3125, 18.3750], [18.3750, 18.3750, 21.5000, 21.5000, 21.5000, 21.5000]]),
    ("channel", None, None, None,
     [[0.0000, 1.3359, 2.0000, 2.6719, 4.0000, 4.6875], [5.8750, 7.3438, 7.3438, 8.8125, 10.2500, 10.2500],
      [11.3125, 13.6250, 13.6250, 15.8750, 15.8750, 15.8750], [18.3750, 18.3750, 21.5000, 21.5000, 21.5000, 21.5000]]),
    ("group", 3, None, None,
     [[0.0000, 1.0703, 1.8750, 2.6719, 4.0000, 4.6875], [6.4375, 7.5000, 7.5000, 8.8125, 10.2500, 10.2500],
      [11.1875, 13.0625, 13.0625, 15.8750, 15.8750, 15.8750], [18.7500, 18.7500, 18.7500, 21.5000, 21.5000, 21.5000]]),
]


@pytest.mark.parametrize("strategy,gs,_mn,_mx,expected", _KA, ids=[k[0] for k in _KA])
def test_known_answer_static_lifecycle(strategy, gs, _mn, _mx, expected):
    w = torch.arange(24, dtype=BF16).reshape(4, 6)
    if strategy == "tensor":
        scale, zp = O.calculate_qparams_minmax(w.reshape(1, -1), num_bits=4, symmetric=True)
        scale, zp = scale.reshape(1), zp.reshape(1)
    else:
        scale, zp = O.calculate_qparams_minmax(w, num_bits=4, group_size=gs, symmetric=True)
    fq = O.fake_quantize(w, scale, zp, num_bits=4, strategy=strategy, group_size=gs)
    assert torch.allclose(fq.float(), torch.tensor(expected, dtype=BF16).float())


@pytest.mark.parametrize("case", cases("qparams"), ids=lambda c: c["key"])
def test_qparams_golden(golden, case):
    t = golden.case("qparams", case["key"])
    scale, zp = O.calculate_qparams_minmax(t["x"], num_bits=case["bits"], group_size=case["group_size"],
                                           symmetric=case["symmetric"])
    assert eq(scale, t["scale"])
    assert eq(zp, t["zp"].to(torch.int8))


# ----------------------------------------------------------------------------- compressors
@pytest.mark.parametrize("case", cases("compressors"), ids=lambda c: c["key"])
def test_compressor_golden(golden, case):
    t = golden.case("compressors", case["key"])
    a = case["args"]
    sd = {k[3:]: v for k, v in t.items() if k.startswith("in.")}
    exp_c = {k[2:]: v for k, v in t.items() if k.startswith("c.")}
    exp_d = {k[2:]: v for k, v in t.items() if k.startswith("d.")}
    if case["format"] == "pack-quantized":
        c = O.pack_quantized_compress(sd, num_bits=a["num_bits"], strategy=a["strategy"],
                                      group_size=a.get("group_size"), symmetric=a["symmetric"])
        assert sorted(c.keys()) == case["compressed_keys"]
        for k in exp_c:
            assert eq(c[k].contiguous(), exp_c[k]), k
        d = O.pack_quantized_decompress(exp_c, num_bits=a["num_bits"], strategy=a["strategy"],
                                        symmetric=a["symmetric"])
        assert sorted(d.keys()) == case["decompressed_keys"]
        for k in exp_d:
            assert eq(d[k].contiguous(), exp_d[k]), k
    else:
        q = O.quantize(sd["weight"], sd["weight_scale"], sd["weight_zero_point"], num_bits=a["num_bits"],
                       strategy=a["strategy"], group_size=a.get("group_size"), dtype=torch.int8)
        assert eq(q, exp_c["weight"])
        d = O.dequantize(exp_c["weight"], exp_c["weight_scale"], exp_c.get("weight_zero_point"))
        assert eq(d, exp_d["weight"])


def _oracle_codec(case, sd, exp_c):
    """the oracle's restatement of one codec call: (compressed dict, decompress function)"""
    a = case["args"]
    if case["format"] == "pack-quantized":
        c = O.pack_quantized_compress(sd, num_bits=a["num_bits"], strategy=a["strategy"], group_size=a.get("group_size"), symmetric=a["symmetric"])
        return c, lambda cc: O.pack_quantized_decompress(cc, num_bits=a["num_bits"], strategy=a["strategy"], symmetric=a["symmetric"])
    w = sd["weight"]
    if a["strategy"] == "block":  # naive_quantized/base.py:72-77: pad to whole blocks, quantize, truncate
        bh, bw = a["block_structure"]
        pr, pc = (-w.shape[0]) % bh, (-w.shape[1]) % bw
        w = torch.nn.functional.pad(w, (0, pc, 0, pr))
    q = O.quantize(w, sd["weight_scale"], sd["weight_zero_point"], num_bits=a["num_bits"], strategy=a["strategy"], group_size=a.get("group_size"),
                   block_structure=a.get("block_structure"), dtype=torch.int8, g_idx=sd.get("weight_g_idx"))
    q = q[: sd["weight"].shape[0], : sd["weight"].shape[1]]
    c = {"weight": q, "weight_scale": sd["weight_scale"]}
    if not a["symmetric"]:
        c["weight_zero_point"] = sd["weight_zero_point"]
    return c, lambda cc: {**{k: v for k, v in cc.items() if k != "weight"},
                          "weight": O.dequantize(cc["weight"], cc["weight_scale"], cc.get("weight_zero_point"), g_idx=cc.get("weight_g_idx"))}


@pytest.mark.parametrize("case", cases("compressors2"), ids=lambda c: c["key"])
def test_compressor_golden_round3(golden, case):
    """activation ordering GROUP / WEIGHT through the class (reference test_pack_quant.py:238-277), 3-D expert weights through
    compress (helpers.py:45-51), channel-symmetric int4, naive int8 `block` with padding (naive_quantized/base.py:72-77)"""
    t = golden.case("compressors2", case["key"])
    sd = {k[3:]: v for k, v in t.items() if k.startswith("in.")}
    exp_c = {k[2:]: v for k, v in t.items() if k.startswith("c.")}
    exp_d = {k[2:]: v for k, v in t.items() if k.startswith("d.")}
    c, dec = _oracle_codec(case, sd, exp_c)
    assert sorted(c.keys()) == case["compressed_keys"]
    for k in exp_c:
        assert eq(c[k].contiguous(), exp_c[k]), k
    if case["round_trip"]:
        d = dec(dict(exp_c))
        assert sorted(d.keys()) == case["decompressed_keys"]
        for k in exp_d:
            assert eq(d[k].contiguous(), exp_d[k]), k


def big_case_inputs(golden, case):
    """the 1024 x 4096 weight of a `compressors_big` case, regenerated from its seed and checked against the recorded digest"""
    import hashlib

    dt = getattr(torch, case["dtype"])
    w = torch.randn(tuple(case["shape"]), generator=torch.Generator().manual_seed(case["seed"])).mul_(0.05).to(dt)
    got = hashlib.sha256(w.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()
    assert got == case["weight_sha256"], "torch.randn no longer reproduces the weight the goldens were generated from: regenerate them (oracle/gen_golden.py compressors_big)"
    sd = {"weight": w}
    sd.update({k[3:]: v for k, v in golden.case("compressors_big", case["key"]).items() if k.startswith("in.")})
    return sd


def digest_matches(t, rec):
    import hashlib

    t = t.detach().cpu().contiguous()
    return (list(t.shape) == rec["shape"] and str(t.dtype).split(".")[-1] == rec["dtype"]
            and hashlib.sha256(t.view(torch.uint8).numpy().tobytes()).hexdigest() == rec["sha256"])


@pytest.mark.parametrize("case", cases("compressors_big"), ids=lambda c: c["key"])
def test_compressor_golden_big(golden, case):
    """1024 x 4096 per format: the oracle against sha256 digests of the reference's outputs"""
    sd = big_case_inputs(golden, case)
    c, dec = _oracle_codec(case, sd, None)
    assert sorted(c.keys()) == sorted(case["compressed"])
    for k, rec in case["compressed"].items():
        assert digest_matches(c[k], rec), k
    d = dec(dict(c))
    for k, rec in case["decompressed"].items():
        assert digest_matches(d[k], rec), k


# ----------------------------------------------------------------------------- sparse primitives
@pytest.mark.parametrize("case", [c for c in cases("sparse") if c["kind"] == "bitmask"], ids=lambda c: c["key"])
def test_bitmask_primitives_golden(golden, case):
    t = golden.case("sparse", case["key"])
    assert eq(O.pack_bitmasks(t["mask"].bool()), t["packed"])
    assert torch.equal(O.unpack_bitmasks(t["packed"], case["shape"]), t["mask"].bool())


def test_bitmask_known_answer():
    # pack_bitmasks([[1,0,0,0,0,0,0,0,0,1]]) == [[1,2]] (verified on the reference)
    m = torch.tensor([[1, 0, 0, 0, 0, 0, 0, 0, 0, 1]], dtype=torch.bool)
    assert O.pack_bitmasks(m).tolist() == [[1, 2]]


@pytest.mark.parametrize("case", [c for c in cases("sparse") if c["kind"] == "cutlass24"], ids=lambda c: c["key"])
def test_cutlass24_golden(golden, case):
    t = golden.case("sparse", case["key"])
    sparse, meta = O.cutlass24_from_dense(t["dense"])
    assert eq(sparse, t["sparse"])
    assert eq(meta, t["meta"])
    assert eq(O.cutlass24_to_dense(t["sparse"], t["meta"]), t["dense_rt"])
    # the oracle's own top-2 mask agrees with the reference's mask_creator when there are no ties
    m = O.sparse24_mask(t["raw"])
    raw = t["raw"].float().abs().reshape(-1, 4)
    no_ties = torch.tensor([len(set(r.tolist())) == 4 for r in raw])
    assert torch.equal(m.reshape(-1, 4)[no_ties], t["mask"].bool().reshape(-1, 4)[no_ties])


@pytest.mark.parametrize("bits", [4, 8])
def test_perm24_golden(golden, bits):
    t = golden.case("sparse", f"perm24_b{bits}")
    assert torch.equal(O.marlin24_perm(bits).to(torch.int64), t["perm"])
    sp, sps = O.marlin24_scale_perms()
    assert sp == t["scale_perm"].tolist() and sps == t["scale_perm_single"].tolist()


# ----------------------------------------------------------------------------- sparse codecs (self-pinned)
@pytest.mark.parametrize("dtype", [BF16, F16, F32, torch.int8])
@pytest.mark.parametrize("shape", [(1, 1), (3, 10), (16, 64), (7, 129), (4, 0)])
def test_bitmask_codec_roundtrip(dtype, shape):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g)
    x = x.masked_fill(torch.rand(shape, generator=g) < 0.5, 0)
    x = (x * 8).to(dtype) if dtype is torch.int8 else x.to(dtype)
    if dtype.is_floating_point and x.numel() > 2:
        x.view(-1)[0] = -0.0
        x.view(-1)[1] = float("nan")
    values, bitmask, row_offsets = O.bitmask_compress(x)
    # cross-check against the reference primitive's definition via pack_bitmasks
    mask = x != 0
    assert torch.equal(bitmask, O.pack_bitmasks(mask).reshape(bitmask.shape))
    counts = mask.reshape(-1, shape[-1]).sum(-1) if shape[-1] else torch.zeros(shape[0], dtype=torch.int64)
    assert torch.equal(row_offsets, torch.cumsum(counts, 0) - counts)
    assert eq(values, x[mask])
    out = O.bitmask_decompress(values, bitmask, shape)
    expect = torch.where(mask, x, torch.zeros_like(x))  # -0.0 comes back as +0.0
    assert eq(out, expect)


@pytest.mark.parametrize("dtype", [BF16, F16, torch.int8])
def test_sparse24_codec_roundtrip(dtype):
    g = torch.Generator().manual_seed(6)
    x = torch.randn((16, 64), generator=g)
    x = (x * 20).to(dtype) if dtype is torch.int8 else x.to(dtype)
    m = O.sparse24_mask(x)
    assert torch.equal(m.reshape(-1, 4).sum(-1), torch.full((x.numel() // 4,), 2))
    pruned = x * m.to(x.dtype)
    values, bitmask = O.sparse24_bitmask_compress(pruned)
    assert values.shape == (16, 32)
    out = O.sparse24_bitmask_decompress(values, bitmask, pruned.shape)
    assert eq(out, torch.where(pruned != 0, pruned, torch.zeros_like(pruned)))


def test_marlin24_pack_matches_torch_restatement():
    """marlin-24 weight packing vs. a direct torch transcription of its definition
    (reshape/permute/gather with the reference's get_permutations_24 table)."""
    g = torch.Generator().manual_seed(8)
    bits = 4
    k, n = 64, 128
    q = torch.randint(0, 16, (k, n), generator=g, dtype=torch.int32)
    perm = O.marlin24_perm(bits).long()
    t = q.reshape(k // 16, 16, n // 16, 16).permute(0, 2, 1, 3).reshape(k // 16, n * 16)
    t = t.reshape(-1, perm.numel())[:, perm].reshape(t.shape)
    pf = 32 // bits
    expect = torch.zeros((t.shape[0], t.shape[1] // pf), dtype=torch.int32)
    for i in range(pf):
        expect |= t[:, i::pf] << (bits * i)
    assert torch.equal(O.marlin24_pack_weights(q, bits), expect)


# ----------------------------------------------------------------------------- FP4 codecs
def test_fp4_primitives_golden(golden):
    t = golden.tensors("fp4")
    for dt in ("torch.float32", "torch.bfloat16", "torch.float16"):
        x, ref = t[f"cast_{dt}.in"], t[f"cast_{dt}.out"]
        got = O.cast_to_fp4(x)
        assert got.dtype == ref.dtype and torch.equal(got, ref)  # values (-0.0 == 0.0 here) ...
        assert torch.equal(torch.signbit(got), torch.signbit(ref))  # ... and the sign of the zeros
    nib = O.fp4_nibbles_of_values(t["pack.in"])
    assert torch.equal(O.pack_fp4(nib), t["pack.out"])
    assert torch.equal(O.fp4_values(O.unpack_fp4(t["pack.out"])), t["unpack.out"])
    assert torch.equal(O.e8m0_encode(t["e8m0.in"]), t["e8m0.out"]) and torch.equal(O.e8m0_decode(t["e8m0.out"]), t["e8m0.back"])


@pytest.mark.parametrize("case", cases("fp4"), ids=lambda c: c["key"])
def test_fp4_compressors_golden(golden, case):
    t = golden.case("fp4", case["key"])
    fmt = case["format"]
    c = O.fp4_compress(t["in.weight"], t["in.weight_scale"], t.get("in.weight_global_scale"), fmt=fmt)
    assert sorted(c) == case["compressed_keys"]
    assert torch.equal(c["weight_packed"], t["comp.weight_packed"])
    assert torch.equal(c["weight_scale"].view(torch.uint8), t["comp.weight_scale"])
    d = O.fp4_decompress(c, fmt=fmt)
    assert sorted(d) == case["decompressed_keys"]
    for name in ("weight", "weight_scale"):
        assert eq(d[name], t[f"dec.{name}"]), name


# ----------------------------------------------------------------------------- FP8 (float-quantized / mxfp8-quantized)
F8 = torch.float8_e4m3fn


def _f8(t):
    """golden fp8 tensors are stored as their bytes"""
    return t.view(F8) if t.dtype == torch.uint8 else t


def eq_f8(a, b):
    """fp8 byte equality; the sign of a NaN is not pinned (torch's own casts disagree between its scalar and
    vectorised paths), every other byte is, -0 included"""
    a, b = a.view(torch.uint8), b.view(torch.uint8)
    an, bn = (a & 0x7F) == 0x7F, (b & 0x7F) == 0x7F
    return a.shape == b.shape and torch.equal(an, bn) and torch.equal(a[~an], b[~bn])


def _fp8_kw(case):
    a = case["args"]
    return dict(num_bits=8, strategy=a["strategy"], group_size=a.get("group_size"), block_structure=a.get("block_structure"), qtype="float")


def test_fp8_cast_matches_torch_for_every_value():
    """the restated float8_e4m3fn cast against torch's own, over every bf16 / fp16 input and a float32 sweep"""
    for dt in (BF16, torch.float16):
        x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dt)
        one = torch.ones((), dtype=dt)
        q = O.quantize(x.reshape(256, 256), one.reshape(1), None, num_bits=8, strategy="tensor", dtype=F8, qtype="float")
        ref = torch.clamp(x / one, -448.0, 448.0).to(F8).reshape(256, 256)
        assert eq_f8(q, ref)
        back = O.dequantize(ref, one.reshape(1), None, strategy="tensor")
        assert eq(back, ref.to(dt))
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(1 << 16, generator=g) * 100, torch.randn(1 << 16, generator=g) * 0.01,
                   torch.arange(0, 520, dtype=torch.float32) * 2.0 ** -9 * 0.5, torch.arange(0, 4096, dtype=torch.float32) * 0.125])
    x = x[: (x.numel() // 64) * 64].reshape(-1, 64)
    q = O.quantize(x, torch.ones(1), None, num_bits=8, strategy="tensor", dtype=F8, qtype="float")
    assert eq_f8(q, torch.clamp(x, -448.0, 448.0).to(F8))


@pytest.mark.parametrize("case", cases("fp8"), ids=lambda c: c["key"])
def test_fp8_quant_golden(golden, case):
    t = golden.case("fp8", case["key"])
    kw = _fp8_kw(case)
    zp = _f8(t["zp"]) if case["zp_dtype"] == "float8_e4m3fn" else t["zp"]
    assert eq_f8(O.quantize(t["x"], t["scale"], zp, dtype=F8, **kw), t["q"])
    assert eq_f8(O.quantize(t["x"], t["scale"], None, dtype=F8, **kw), t["q_nozp"])
    assert eq(O.quantize(t["x"], t["scale"], zp, **kw), t["qf"])
    assert eq(O.fake_quantize(t["x"], t["scale"], zp, **kw), t["fq"])
    dkw = {k: v for k, v in kw.items() if k not in ("num_bits", "qtype")}
    assert eq(O.dequantize(_f8(t["q"]), t["scale"], zp, **dkw), t["dq"])
    if kw["strategy"] != "block":
        assert eq(O.dequantize(_f8(t["q"]), t["scale"], zp), t["dq_inferred"])


@pytest.mark.parametrize("case", cases("fp8", "codecs"), ids=lambda c: c["key"])
def test_fp8_codecs_golden(golden, case):
    """float-quantized / naive-quantized(float) (naive_quantized/base.py:48-126) and mxfp8-quantized (mxfp8/base.py:47-101)"""
    t = golden.case("fp8", case["key"])
    a = case["args"]
    sd = {k[3:]: v for k, v in t.items() if k.startswith("in.")}
    exp_c = {k[2:]: v for k, v in t.items() if k.startswith("c.")}
    exp_d = {k[2:]: v for k, v in t.items() if k.startswith("d.")}
    q = O.quantize(sd["weight"], sd["weight_scale"], _f8(sd["weight_zero_point"]), num_bits=8, strategy=a["strategy"],
                   group_size=a.get("group_size"), block_structure=a.get("block_structure"), dtype=F8, qtype="float")
    assert eq_f8(q, exp_c["weight"])
    assert "weight_zero_point" not in exp_c  # symmetric: dropped
    if case["format"] == "mxfp8-quantized":
        assert torch.equal(O.e8m0_encode(sd["weight_scale"]), exp_c["weight_scale"])
        scale = O.e8m0_decode(exp_c["weight_scale"])
        assert eq(scale, exp_d["weight_scale"])
    else:
        scale = exp_c["weight_scale"]
        assert eq(scale, sd["weight_scale"])
    # decompress never sees the args: the strategy is inferred from the scale shape (forward.py:99-130), which for a
    # padded block layout (300 x 400 under 128 x 128 blocks -> scale 3 x 4) means 100 x 100 blocks, as upstream
    d = O.dequantize(_f8(exp_c["weight"]), scale, None)
    assert eq(d, exp_d["weight"])


# ----------------------------------------------------------------------------- FLOAT 4-bit quantize / dequantize / fake_quantize
def _fp4q_inputs(t, case):
    zp = _f8(t["zp"]) if case["zp_dtype"] == "float8_e4m3fn" else t["zp"]
    a = case["args"]
    kw = dict(num_bits=4, strategy=a["strategy"], group_size=a.get("group_size"), qtype="float", global_scale=t.get("gs"))
    return zp, kw


@pytest.mark.parametrize("case", cases("fp4q"), ids=lambda c: c["key"])
def test_fp4_quant_golden(golden, case):
    """quantize / fake_quantize / dequantize with FLOAT 4-bit args, with and without a global scale"""
    t = golden.case("fp4q", case["key"])
    zp, kw = _fp4q_inputs(t, case)
    assert eq(O.quantize(t["x"], t["scale"], zp, **kw), t["qf"])
    assert eq(O.quantize(t["x"], t["scale"], None, **kw), t["qf_nozp"])
    assert eq(O.fake_quantize(t["x"], t["scale"], zp, **kw), t["fq"])
    dkw = {k: v for k, v in kw.items() if k not in ("num_bits", "qtype")}
    assert eq(O.dequantize(t["qf"], t["scale"], zp, **dkw), t["dq"])


# ----------------------------------------------------------------------------- qparams of the FLOAT schemes
def _qpf_kind(case):
    return "fp8" if case["kind"].startswith("fp8") else case["kind"]


@pytest.mark.parametrize("case", cases("qparams_float"), ids=lambda c: c["key"])
def test_qparams_float_golden(golden, case):
    t = golden.case("qparams_float", case["key"])
    s = O.calculate_qparams_float(t["x"], kind=_qpf_kind(case), group_size=case["group_size"], global_scale=t.get("gs"))
    assert eq(s, t["scale"])
    assert not bool(t["zp"].view(torch.uint8).any())  # symmetric: all-zero zero points of the scheme's zp_dtype
    if "gs" in t:
        assert eq(O.generate_gparam(t["x"]), t["gs"])


# ----------------------------------------------------------------------------- torch-eager restatement (the CPU baseline of bench.py)
@pytest.mark.parametrize("case", cases("pack"), ids=lambda c: c["key"])
def test_eager_ref_pack_unpack_golden(golden, case):
    """oracle/eager_ref.py (the reference's op sequence, timed as bench.py's cpu_baseline) against the reference-generated goldens"""
    import eager_ref as E

    t = golden.case("pack", case["key"])
    bits, pd = case["bits"], case["packed_dim"]
    assert eq(E.pack_to_int32(t["value"], bits, packed_dim=pd).contiguous(), t["packed"])
    if not case.get("oob"):
        assert eq(E.unpack_from_int32(t["packed"], bits, torch.Size(case["shape"]), packed_dim=pd).contiguous(), t["value"])


@pytest.mark.parametrize("dtype", [BF16, F16, F32])
@pytest.mark.parametrize("bits,strategy,gs,sym", [(4, "group", 128, True), (4, "group", 32, False), (4, "channel", None, True),
                                                  (8, "tensor", None, True), (8, "channel", None, False), (3, "group", 64, False)])
def test_eager_ref_matches_the_oracle(dtype, bits, strategy, gs, sym):
    import eager_ref as E

    g = torch.Generator().manual_seed(bits * 7 + (gs or 0))
    w = torch.randn(96, 256, generator=g).to(dtype)
    w[0, :8] = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e-8, -1e-8, 3.0]).to(dtype)
    finite = torch.nan_to_num(w, nan=0.0, posinf=4.0, neginf=-4.0)  # the observer never sees non-finite values
    scale, zp = O.calculate_qparams_minmax(finite.reshape(1, -1) if strategy == "tensor" else finite, num_bits=bits,
                                           group_size=gs, symmetric=sym)
    if strategy == "tensor":
        scale, zp = scale.reshape(1), zp.reshape(1)
    q_e = E.quantize(w, scale, zp, num_bits=bits, strategy=strategy, group_size=gs)
    q_o = O.quantize(w, scale, zp, num_bits=bits, strategy=strategy, group_size=gs, dtype=torch.int8)
    assert eq(q_e, q_o)
    assert eq(E.dequantize(q_e, scale, zp), O.dequantize(q_o, scale, zp))
    sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
    if strategy != "tensor":
        c_e = E.pack_quantized_compress(sd, num_bits=bits, strategy=strategy, group_size=gs, symmetric=sym)
        c_o = O.pack_quantized_compress(sd, num_bits=bits, strategy=strategy, group_size=gs, symmetric=sym)
        assert sorted(c_e) == sorted(c_o)
        assert all(eq(c_e[k].contiguous(), c_o[k].contiguous()) for k in c_o)
        d_e = E.pack_quantized_decompress(c_e, num_bits=bits, strategy=strategy, symmetric=sym)
        d_o = O.pack_quantized_decompress(c_o, num_bits=bits, strategy=strategy, symmetric=sym)
        assert eq(d_e["weight"], d_o["weight"])


def test_eager_ref_matches_the_reference_itself():
    """where the reference is importable (the build container): same state dicts from PackedQuantizationCompressor / IntQuantizationCompressor"""
    import ref_import

    if not ref_import.available():
        pytest.skip("upstream reference sources not present on this machine")
    import eager_ref as E

    ref_import.import_reference()
    from compressed_tensors.compressors.naive_quantized.base import IntQuantizationCompressor as RI
    from compressed_tensors.compressors.pack_quantized.base import PackedQuantizationCompressor as RP
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme

    torch.manual_seed(3)
    w = torch.randn(128, 512, dtype=BF16)
    for sym in (True, False):
        scale, zp = O.calculate_qparams_minmax(w, num_bits=4, group_size=128, symmetric=sym)
        sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
        sch = QuantizationScheme(targets=["Linear"], weights=QuantizationArgs(num_bits=4, group_size=128, symmetric=sym, strategy="group"))
        rc = RP.compress(sd, sch)
        ec = E.pack_quantized_compress(sd, num_bits=4, strategy="group", group_size=128, symmetric=sym)
        assert sorted(rc) == sorted(ec) and all(eq(rc[k].contiguous(), ec[k].contiguous()) for k in rc)
        rd = RP.decompress(rc, sch)
        ed = E.pack_quantized_decompress(ec, num_bits=4, strategy="group", symmetric=sym)
        assert sorted(rd) == sorted(ed) and all(eq(rd[k].contiguous(), ed[k].contiguous()) for k in rd)
    s1 = (w.abs().max().float() / 127).to(BF16).reshape(1)
    z1 = torch.zeros(1, dtype=torch.int8)
    a8 = QuantizationArgs(num_bits=8, strategy="tensor", symmetric=True)
    sch8 = QuantizationScheme(targets=["Linear"], weights=a8, input_activations=a8)
    sd8 = {"weight": w, "weight_scale": s1, "weight_zero_point": z1}
    rc, ec = RI.compress(sd8, sch8), E.int_quantized_compress(sd8)
    assert sorted(rc) == sorted(ec) and all(eq(rc[k], ec[k]) for k in rc)
    assert eq(RI.decompress(rc, sch8)["weight"], E.int_quantized_decompress(ec)["weight"])
