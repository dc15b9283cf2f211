"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's golden
vectors.  Every test here needs an MI355X:  python -m pytest tests -m gpu"""
import math
import os
import re

import pytest
import torch
from _golden import cases
from test_oracle_golden import _kw, eq, eq_f8

import oracle as O

pytestmark = pytest.mark.gpu

BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cta():
    import compressed_tensors_amd as m
    from compressed_tensors_amd import _lib

    _lib.load()  # fail loudly if the HIP extension is missing
    return m


def d(t, dev):
    return None if t is None else t.to(dev)


def special_values(dtype):
    v = [float("nan"), float("inf"), -float("inf"), 0.0, -0.0, 1e30, -1e30, 1e-30, 2.498, 3.496, 0.5, 1.5, 2.5, -0.5,
         -1.5, -2.5, 6.5, 7.5, -7.5, -8.5, 127.5, -128.5, 65504.0, 1e-8]
    return torch.tensor(v, dtype=torch.float32).to(dtype)


# ----------------------------------------------------------------------------- golden: pack / unpack
@pytest.mark.parametrize("case", cases("pack"), ids=lambda c: c["key"])
def test_pack_unpack_golden(golden, cta, dev, case):
    t = golden.case("pack", case["key"])
    bits, pd = case["bits"], case["packed_dim"]
    packed = cta.codec.pack_to_int32(d(t["value"], dev), bits, packed_dim=pd)
    assert packed.is_cuda and packed.dtype == torch.int32 and packed.is_contiguous()
    assert eq(packed.cpu(), t["packed"])
    if not case.get("oob"):
        un = cta.codec.unpack_from_int32(d(t["packed"], dev), bits, torch.Size(case["shape"]), packed_dim=pd)
        assert eq(un.cpu(), t["value"])


@pytest.mark.parametrize("bits", [1, 2, 4, 8])
@pytest.mark.parametrize("k", [33, 64, 100, 1024])
def test_old_format_compat(cta, dev, bits, k):
    """reference tests/test_compressors/test_pack_quant.py:386-416"""
    from test_oracle_golden import _old_pack

    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    v = torch.randint(lo, hi + 1, (64, k), dtype=torch.int8)
    old = _old_pack(v, bits)
    assert torch.equal(cta.codec.pack_to_int32(v.to(dev), bits).cpu(), old)
    assert torch.equal(cta.codec.unpack_from_int32(old.to(dev), bits, torch.Size((64, k))).cpu(), v)


@pytest.mark.parametrize("bits", range(1, 9))
@pytest.mark.parametrize("shape", [(256, 1024), (512, 100), (128, 33), (64, 4096), (3, 8, 40)])
def test_pack_unpack_vs_oracle(cta, dev, bits, shape):
    g = torch.Generator().manual_seed(bits * 100 + shape[-1])
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    v = torch.randint(lo, hi + 1, shape, dtype=torch.int8, generator=g)
    packed = cta.codec.pack_to_int32(v.to(dev), bits)
    assert packed.shape == (*shape[:-1], math.ceil(shape[-1] * bits / 32))
    assert torch.equal(packed.cpu(), O.pack_to_int32(v, bits).contiguous())
    assert torch.equal(cta.codec.unpack_from_int32(packed, bits, torch.Size(shape)).cpu(), v)


# ----------------------------------------------------------------------------- golden: quantization
@pytest.mark.parametrize("case", cases("quant"), ids=lambda c: c["key"])
def test_quant_golden(golden, cta, dev, case):
    t = golden.case("quant", case["key"])
    kw = _kw(case)
    x, s, z = d(t["x"], dev), d(t["scale"], dev), d(t["zp"], dev)
    g_idx = d(t.get("g_idx"), dev)
    q8 = cta.codec.quantize_tensor(x, s, z, dtype=torch.int8, g_idx=g_idx, **kw)
    assert eq(q8.cpu(), t["q8"])
    assert eq(cta.codec.fake_quantize_tensor(x, s, z, g_idx=g_idx, **kw).cpu(), t["fq"])
    dkw = dict(kw)
    bits = dkw.pop("num_bits")
    assert eq(cta.codec.dequantize_tensor(d(t["q8"], dev), s, z, g_idx=g_idx, **dkw).cpu(), t["dq"])
    if "qf" in t:
        assert eq(cta.codec.quantize_tensor(x, s, z, g_idx=g_idx, **kw).cpu(), t["qf"])
        assert eq(cta.codec.quantize_tensor(x, s, None, dtype=torch.int8, **kw).cpu(), t["q8_nozp"])
        if kw["strategy"] != "block":
            assert eq(cta.codec.dequantize_tensor(d(t["q8"], dev), s, z).cpu(), t["dq_inferred"])
    # fused kernels against the unfused golden results
    packed = cta.codec.quantize_and_pack(x, s, z, g_idx=g_idx, **kw)
    assert eq(packed.cpu(), O.pack_to_int32(t["q8"], bits).contiguous())
    out = cta.codec.unpack_and_dequantize(packed, t["x"].shape, s, z, num_bits=bits, g_idx=g_idx, **dkw)
    assert eq(out.cpu(), t["dq"])


@pytest.mark.parametrize("case", cases("qparams"), ids=lambda c: c["key"])
def test_qparams_golden(golden, cta, dev, case):
    t = golden.case("qparams", case["key"])
    scale, zp = cta.codec.minmax_qparams(d(t["x"], dev), num_bits=case["bits"], group_size=case["group_size"],
                                         symmetric=case["symmetric"])
    assert eq(scale.cpu(), t["scale"])
    assert eq(zp.cpu(), t["zp"].to(torch.int8))


# ----------------------------------------------------------------------------- golden: compressors
def _scheme(cta, case):
    a = case["args"]
    args = cta.QuantizationArgs(**a)
    act = cta.QuantizationArgs(num_bits=8, strategy="tensor", symmetric=True) if case["format"] == "int-quantized" else None
    return cta.QuantizationScheme(targets=["Linear"], weights=args, input_activations=act)


@pytest.mark.parametrize("case", cases("compressors"), ids=lambda c: c["key"])
def test_compressor_golden(golden, cta, dev, case):
    t = golden.case("compressors", case["key"])
    sd = {k[3:]: d(v, dev) for k, v in t.items() if k.startswith("in.")}
    exp_c = {k[2:]: v for k, v in t.items() if k.startswith("c.")}
    exp_d = {k[2:]: v for k, v in t.items() if k.startswith("d.")}
    scheme = _scheme(cta, case)
    comp = cta.BaseCompressor.get_value_from_registry(case["format"])
    sd_before = dict(sd)
    c = comp.compress(sd, scheme)
    assert sd == sd_before, "compress must not mutate its input dict"
    assert sorted(c.keys()) == case["compressed_keys"]
    for k, v in exp_c.items():
        assert eq(c[k].cpu().contiguous(), v), k
        if k == "weight_shape":
            assert c[k].device.type == "cpu" and c[k].dtype == torch.int64
        else:
            assert c[k].is_cuda, k
    assert c["weight_scale"] is sd["weight_scale"], "untouched tensors are returned by identity"
    dd = comp.decompress({k: d(v, dev) if k != "weight_shape" else v for k, v in exp_c.items()}, scheme)
    assert sorted(dd.keys()) == case["decompressed_keys"]
    for k, v in exp_d.items():
        assert eq(dd[k].cpu().contiguous(), v), k
    assert sorted(comp.compression_param_names(scheme)) == sorted(k for k in case["compressed_keys"])


@pytest.mark.parametrize("case", cases("compressors2"), ids=lambda c: c["key"])
def test_compressor_golden_round3(golden, cta, dev, case):
    """reference-generated class-level vectors the first family lacked: activation ordering GROUP / WEIGHT
    (reference test_pack_quant.py:238-277), 3-D expert weights through compress (helpers.py:45-51), channel-symmetric int4,
    naive int8 `block` with padding (naive_quantized/base.py:72-77)"""
    t = golden.case("compressors2", case["key"])
    sd = {k[3:]: d(v, dev) for k, v in t.items() if k.startswith("in.")}
    exp_c = {k[2:]: v for k, v in t.items() if k.startswith("c.")}
    exp_d = {k[2:]: v for k, v in t.items() if k.startswith("d.")}
    scheme = _scheme(cta, case)
    comp = cta.BaseCompressor.get_value_from_registry(case["format"])
    c = comp.compress(dict(sd), scheme)
    assert sorted(c.keys()) == case["compressed_keys"]
    for k, v in exp_c.items():
        assert eq(c[k].cpu().contiguous(), v), k
    if case["round_trip"]:
        dd = comp.decompress({k: d(v, dev) if k != "weight_shape" else v for k, v in exp_c.items()}, scheme)
        assert sorted(dd.keys()) == case["decompressed_keys"]
        for k, v in exp_d.items():
            assert eq(dd[k].cpu().contiguous(), v), k


@pytest.mark.parametrize("case", cases("compressors_big"), ids=lambda c: c["key"])
def test_compressor_golden_big(golden, cta, dev, case):
    """1024 x 4096 per format / scheme — the sizes at which the flat / lean / rows-per-workgroup kernels are selected — against
    sha256 digests of the reference's own outputs (weight regenerated from its seed and checked against the recorded digest)"""
    from test_oracle_golden import big_case_inputs, digest_matches

    sd = {k: d(v, dev) for k, v in big_case_inputs(golden, case).items()}
    scheme = _scheme(cta, case)
    comp = cta.BaseCompressor.get_value_from_registry(case["format"])
    c = comp.compress(dict(sd), scheme)
    assert sorted(c.keys()) == sorted(case["compressed"])
    for k, rec in case["compressed"].items():
        assert digest_matches(c[k], rec), f"compressed[{k}] differs from the reference"
    dd = comp.decompress(dict(c), scheme)
    for k, rec in case["decompressed"].items():
        assert digest_matches(dd[k], rec), f"decompressed[{k}] differs from the reference"


@pytest.mark.parametrize("dtype", [BF16, F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("bits", [1, 2, 3, 5, 6, 7])
def test_lean_kernels_of_the_other_bit_widths_vs_oracle(cta, dev, bits, dtype):
    """Round 6 (VERDICT r05 missing #3): the widths next to 4 and 8 (helpers.py:39-42 allows 1..8; the reference's own
    test_pack_quant.py:143-146 parametrises them) in the common checkpoint layout take the lean kernels of ct_quant_wb.hip — one pack group
    per lane on the compress side, one 8-element unit per lane with a wave-wide word exchange on the decompress side.  Packed words and
    decompressed weights bit for bit against the CPU oracle: symmetric and asymmetric, groups of 32 / 64 / 128 / 96 (not a power of two) /
    the whole row, unit counts that are not a multiple of 64 (partial wave windows), scales outside the reciprocal fast path, inf / NaN
    weights, every code of the width in every position of a group."""
    gen = torch.Generator().manual_seed(100 * bits + (1 if dtype is F16 else 0))
    half = 1 << (bits - 1)
    for (rows, cols, group, sym) in [(64, 1024, 128, True), (64, 1024, 128, False), (33, 384, 96, False), (7, 96, 32, True), (130, 2048, 64, False),
                                     (5, 4096, None, True), (1, 32, 32, False), (256, 512, 128, True)]:
        w = torch.randn(rows, cols, generator=gen)
        w[0, :5] = torch.tensor([float("inf"), float("-inf"), float("nan"), 0.0, -0.0])
        w = w.to(dtype)
        s, z = O.calculate_qparams_minmax(w.masked_fill(~torch.isfinite(w.float()), 0), num_bits=bits, group_size=group, symmetric=sym)
        if rows >= 5:  # scales outside the fast-path ranges (exact-division path), a zero point at each end
            s[1, 0], s[2, 0] = torch.tensor(2.0 ** -20).to(dtype), torch.tensor(3.0e4).to(dtype)
            if not sym:
                z[3, 0], z[4, 0] = -half, half - 1
        kw = dict(num_bits=bits, strategy="group" if group else "channel", group_size=group)
        # every code in every position: w = code * scale exactly representable for small codes; random elsewhere
        sd = {"weight": w, "weight_scale": s, "weight_zero_point": z}
        ref_c = O.pack_quantized_compress(sd, symmetric=sym, **kw)
        got = cta.codec.quantize_and_pack(w.to(dev), s.to(dev), z.to(dev), **kw)
        assert got.shape == ref_c["weight_packed"].shape and torch.equal(got.cpu(), ref_c["weight_packed"]), (rows, cols, group, sym)
        back = cta.codec.unpack_and_dequantize(got, (rows, cols), s.to(dev), None if sym else z.to(dev), num_bits=bits)
        ref_d = O.pack_quantized_decompress(ref_c, num_bits=bits, strategy=kw["strategy"], symmetric=sym)
        assert eq(back.cpu(), ref_d["weight"]), (rows, cols, group, sym)
    # every code of the width at every position of a pack group, against the packer itself
    q = ((torch.arange(32 * (1 << bits)) // 32 + torch.arange(32 * (1 << bits)) % 32) % (1 << bits) - half).to(torch.int8).reshape(-1, 32).repeat(1, 4).contiguous()
    one = torch.ones(q.shape[0], 1, dtype=dtype)
    words = O.pack_to_int32(q, bits).contiguous()
    back = cta.codec.unpack_and_dequantize(words.to(dev), q.shape, one.to(dev), None, num_bits=bits)
    assert torch.equal(back.cpu().float(), q.float())
    again = cta.codec.quantize_and_pack(q.to(dtype).to(dev), one.to(dev), None, num_bits=bits, strategy="channel")
    assert torch.equal(again.cpu(), words)


@pytest.mark.parametrize("bits", [2, 3, 6])
def test_lean_other_widths_full_size(cta, dev, bits):
    """8192 x 8192 bf16 g128 (the size the roofline row is quoted on): compress on the device equals the oracle on a 256-row slice and
    the composition quantize -> pack_to_int32 on the whole tensor; decompress(compress(W)) == fake_quantize(W)"""
    n = 8192
    w = torch.randn(n, n, dtype=torch.float32, device=dev, generator=torch.Generator(device=dev).manual_seed(bits)).to(BF16)
    for sym in (True, False):
        s, z = cta.codec.minmax_qparams(w, num_bits=bits, group_size=128, symmetric=sym)
        kw = dict(num_bits=bits, strategy="group", group_size=128)
        packed = cta.codec.quantize_and_pack(w, s, z, **kw)
        q = cta.codec.quantize_tensor(w, s, z, dtype=torch.int8, **kw)
        assert torch.equal(packed, cta.codec.pack_to_int32(q, bits))
        sl = slice(4000, 4256)
        ref = O.pack_quantized_compress({"weight": w[sl].cpu(), "weight_scale": s[sl].cpu(), "weight_zero_point": z[sl].cpu()}, symmetric=sym, **kw)
        assert torch.equal(packed[sl].cpu(), ref["weight_packed"])
        back = cta.codec.unpack_and_dequantize(packed, (n, n), s, None if sym else z, num_bits=bits)
        assert torch.equal(back, cta.codec.fake_quantize_tensor(w, s, z, **kw))
        assert eq(back[sl].cpu(), O.pack_quantized_decompress(ref, num_bits=bits, strategy="group", symmetric=sym)["weight"])


def test_fuzz_parity_bounded():
    """tools/fuzz_parity.py (every public codec entry point against the oracle on random shapes / dtypes / strategies / special
    values) with a fixed seed and a bounded number of cases, so that the driver's -m gpu run re-runs it"""
    import subprocess
    import sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "90", "20260926", "1500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    m = re.search(r"fuzz: (\d+) random cases, no mismatch", r.stdout)
    assert m and int(m.group(1)) >= 400, r.stdout[-500:]


# ----------------------------------------------------------------------------- random vs oracle
QCASES = [
    # dtype, scale dtype, bits, strategy, group, shape, symmetric
    (BF16, BF16, 4, "group", 128, (256, 4096), True),
    (BF16, BF16, 4, "group", 128, (256, 4096), False),
    (BF16, BF16, 4, "group", 32, (64, 256), False),
    (BF16, BF16, 4, "channel", None, (128, 1000), False),
    (BF16, BF16, 8, "channel", None, (128, 1024), True),
    (BF16, F32, 4, "group", 128, (64, 1024), False),
    (F16, F16, 4, "group", 128, (128, 2048), False),
    (F16, F16, 8, "tensor", None, (64, 512), True),
    (F32, F32, 4, "group", 64, (64, 512), False),
    (F32, F32, 3, "channel", None, (33, 70), False),
    (F32, F32, 4, "group", 128, (96, 1024), True),      # the flat fp32 kernels (W4 words, quads)
    (F32, F32, 4, "group", 128, (96, 1024), False),
    (F32, BF16, 4, "group", 32, (64, 512), False),
    (F32, F32, 8, "channel", None, (64, 2048), False),
    (F32, F32, 8, "tensor", None, (64, 520), True),
    (F32, F32, 6, "block", None, (64, 256), True),
    (BF16, BF16, 5, "group", 32, (36, 96), False),
    (BF16, BF16, 2, "group", 16, (17, 48), True),
    (BF16, BF16, 1, "channel", None, (8, 100), True),
    (BF16, BF16, 7, "tensor", None, (16, 99), False),
    (BF16, BF16, 8, "block", None, (64, 256), True),
]


def _make_qcase(xdt, sdt, bits, strategy, gs, shape, sym, seed=0):
    g = torch.Generator().manual_seed(seed + bits)
    x = torch.randn(shape, generator=g).mul(2.0).to(xdt)
    sp = special_values(xdt)
    x.view(-1)[: sp.numel()] = sp
    xf = torch.nan_to_num(x.float(), nan=0.0, posinf=4.0, neginf=-4.0).clamp(-9, 9).to(xdt)
    if strategy == "tensor":
        scale, zp = O.calculate_qparams_minmax(xf.reshape(1, -1), num_bits=bits, symmetric=sym)
        scale, zp = scale.reshape(1), zp.reshape(1)
    elif strategy == "block":
        bs = [8, 64]
        scale = (torch.rand((shape[0] // 8, shape[1] // 64), generator=g) * 0.05 + 0.01).to(xdt)
        zp = torch.zeros(scale.shape, dtype=torch.int8)
    else:
        scale, zp = O.calculate_qparams_minmax(xf, num_bits=bits, group_size=gs, symmetric=sym)
    kw = dict(num_bits=bits, strategy=strategy, group_size=gs, block_structure=[8, 64] if strategy == "block" else None)
    return x, scale.to(sdt), zp, kw


@pytest.mark.parametrize("xdt,sdt,bits,strategy,gs,shape,sym", QCASES)
def test_quant_paths_vs_oracle(cta, dev, xdt, sdt, bits, strategy, gs, shape, sym):
    x, scale, zp, kw = _make_qcase(xdt, sdt, bits, strategy, gs, shape, sym)
    q_ref = O.quantize(x, scale, zp, dtype=torch.int8, **kw)
    q = cta.codec.quantize_tensor(x.to(dev), scale.to(dev), zp.to(dev), dtype=torch.int8, **kw)
    assert eq(q.cpu(), q_ref)
    fq = cta.codec.fake_quantize_tensor(x.to(dev), scale.to(dev), zp.to(dev), **kw)
    assert eq(fq.cpu(), O.fake_quantize(x, scale, zp, **kw))
    dkw = {k: v for k, v in kw.items() if k != "num_bits"}
    dq_ref = O.dequantize(q_ref, scale, zp, **dkw)
    assert eq(cta.codec.dequantize_tensor(q, scale.to(dev), zp.to(dev), **dkw).cpu(), dq_ref)
    packed = cta.codec.quantize_and_pack(x.to(dev), scale.to(dev), zp.to(dev), **kw)
    assert eq(packed.cpu(), O.pack_to_int32(q_ref, bits).contiguous())
    out = cta.codec.unpack_and_dequantize(packed, x.shape, scale.to(dev), zp.to(dev), **kw)
    assert eq(out.cpu(), dq_ref)
    # symmetric call without a zero point takes the no-zp kernel path
    q0 = cta.codec.quantize_and_pack(x.to(dev), scale.to(dev), None, **kw)
    assert eq(q0.cpu(), O.pack_to_int32(O.quantize(x, scale, None, dtype=torch.int8, **kw), bits).contiguous())


@pytest.mark.parametrize("qtype", ["float", "int"])
@pytest.mark.parametrize("dt", [BF16, F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("shape,block,sym", [((384, 2304), [128, 128], True), ((256, 1024), [128, 128], False), ((200, 1200), [24, 40], False),
                                             ((130, 4112), [64, 16], True), ((96, 2304), [32, 768], False)],
                         ids=["128x128_288_units", "128x128_asym", "odd_blocks_ragged", "ragged_rows_16_wide", "wide_blocks"])
def test_block_strategy_flat_kernels_vs_oracle(cta, dev, qtype, dt, shape, block, sym):
    """FP8 / int8 block quantization (the scale of element (r, c) is scale[r // bh][c // bw], forward.py:198-216) on the flat 8-bit kernels: their scale
    index is three 32-bit multiply-highs (w4_scale_index, nf_fast); rows that are not a power-of-two number of units, ragged last blocks, zero points"""
    r, c = shape
    bh, bw = block
    g = torch.Generator().manual_seed(r + c)
    x = torch.randn(shape, generator=g).mul(3.0).to(dt)
    sp = special_values(dt)
    x.view(-1)[: sp.numel()] = sp
    scale = (torch.rand((-(-r // bh), -(-c // bw)), generator=g) * 0.05 + 0.01).to(dt)
    zp = None if sym else torch.randint(-9, 9, scale.shape, generator=g, dtype=torch.int8)
    odt = torch.float8_e4m3fn if qtype == "float" else torch.int8
    kw = dict(num_bits=8, strategy="block", block_structure=block, qtype=qtype)
    q_ref = O.quantize(x, scale, zp, dtype=odt, **kw)
    q = cta.codec.quantize_tensor(x.to(dev), scale.to(dev), None if sym else zp.to(dev), dtype=odt, **kw)
    assert eq_f8(q.cpu(), q_ref) if qtype == "float" else eq(q.cpu(), q_ref)
    dkw = dict(strategy="block", block_structure=block)
    back = cta.codec.dequantize_tensor(q, scale.to(dev), None if sym else zp.to(dev), **dkw)
    assert eq(back.cpu(), O.dequantize(q_ref, scale, zp, **dkw))
    fq = cta.codec.fake_quantize_tensor(x.to(dev), scale.to(dev), None if sym else zp.to(dev), **kw)
    assert eq(fq.cpu(), O.fake_quantize(x, scale, zp, **kw))


@pytest.mark.parametrize("dt", [BF16, F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("sym", [True, False], ids=["sym", "asym"])
@pytest.mark.parametrize("shape,gs", [((32, 512), 128), ((96, 4096), 128), ((7, 1032), 8), ((16, 8192 * 2), 16), ((5, 8288), 32), ((9, 4128), 16)],
                         ids=["small", "wide", "ragged_chunk", "1024_groups", "second_chunk_tail", "half_chunk_tail"])
def test_gidx_vs_oracle(cta, dev, dt, sym, shape, gs):
    """activation ordering (weight_g_idx): quantize, the packed words and the decompressed weight against the oracle — the flat W4
    g_idx kernels (a workgroup inside one row, the row's scales in LDS) for 16-bit weights, special values included"""
    g = torch.Generator().manual_seed(11 + shape[1])
    rows, cols = shape
    x = torch.randn(shape, generator=g).to(dt)
    sp = special_values(dt)
    x.view(-1)[: sp.numel()] = sp
    perm = torch.randperm(cols, generator=g)
    g_idx = (torch.arange(cols, dtype=torch.int32) // gs)[perm].contiguous()
    scale = (torch.rand((rows, cols // gs), generator=g) * 0.3 + 0.05).to(dt)
    scale[0, 0] = 2.0 ** -30 if dt == BF16 else 2.0 ** -15  # one scale outside the reciprocal range of fp16
    zp = torch.zeros(rows, cols // gs, dtype=torch.int8) if sym else torch.randint(-8, 8, (rows, cols // gs), generator=g).to(torch.int8)
    kw = dict(num_bits=4, strategy="group", group_size=gs)
    ref = O.quantize(x, scale, zp, dtype=torch.int8, g_idx=g_idx, **kw)
    got = cta.codec.quantize_tensor(x.to(dev), scale.to(dev), zp.to(dev), dtype=torch.int8, g_idx=g_idx.to(dev), **kw)
    assert eq(got.cpu(), ref)
    packed = cta.codec.quantize_and_pack(x.to(dev), scale.to(dev), None if sym else zp.to(dev), g_idx=g_idx.to(dev), **kw)
    assert eq(packed.cpu(), O.pack_to_int32(ref, 4).contiguous())
    out = cta.codec.unpack_and_dequantize(packed, x.shape, scale.to(dev), None if sym else zp.to(dev), g_idx=g_idx.to(dev), **kw)
    assert eq(out.cpu(), O.dequantize(ref, scale, zp, strategy="group", group_size=gs, g_idx=g_idx))


def test_gidx_rewritten_in_place_is_seen(cta, dev):
    """a weight_g_idx first seen as all -1 (plain column order) and then filled through `param.data.copy_` — which
    neither moves the pointer nor bumps the version counter — must use the new ordering on the next call"""
    g = torch.Generator().manual_seed(5)
    rows, cols, gs = 16, 1024, 128
    x = torch.randn(rows, cols, generator=g).to(BF16)
    scale = (torch.rand((rows, cols // gs), generator=g) * 0.3 + 0.05).to(BF16)
    kw = dict(num_bits=4, strategy="group", group_size=gs)
    param = torch.nn.Parameter(torch.full((cols,), -1, dtype=torch.int32, device=dev), requires_grad=False)
    plain = cta.codec.quantize_and_pack(x.to(dev), scale.to(dev), None, g_idx=param.data, **kw)
    assert eq(plain.cpu(), O.pack_to_int32(O.quantize(x, scale, None, dtype=torch.int8, **kw), 4).contiguous())
    g_idx = (torch.arange(cols, dtype=torch.int32) // gs)[torch.randperm(cols, generator=g)].contiguous()
    param.data.copy_(g_idx.to(dev))
    ordered = cta.codec.quantize_and_pack(x.to(dev), scale.to(dev), None, g_idx=param.data, **kw)
    ref = O.quantize(x, scale, None, dtype=torch.int8, g_idx=g_idx, **kw)
    assert eq(ordered.cpu(), O.pack_to_int32(ref, 4).contiguous())
    out = cta.codec.unpack_and_dequantize(ordered, x.shape, scale.to(dev), None, g_idx=param.data, **kw)
    assert eq(out.cpu(), O.dequantize(ref, scale, None, strategy="group", group_size=gs, g_idx=g_idx))


def test_bf16_reciprocal_fast_path_is_exact(cta, dev):
    """exhaustive over all 65536 x 65536 bf16 (x, scale) pairs inside the fast-path range"""
    assert cta.codec.selftest_bf16_div(0, 65536) == 0


def test_f16_newton_quotient_is_exact(cta, dev):
    """exhaustive: every fp16 x against every fp16 scale of the fast-path range (marlin-24 front end)"""
    assert cta.codec.selftest_f16_div(0, 65536) == 0


def test_fp32_quotient_ties(cta, dev):
    """the fp32 quantize paths multiply by a reciprocal and divide only when the clamped value is too close to a half-integer to call:
    feed them exact ties (x = (k + 0.5) s for scales with a short significand), the neighbours one and two ulps either side, huge,
    tiny, infinite, NaN and signed-zero inputs, scales from 2^-100 to 2^100 and outside; every code, fake-quantized value and
    packed word must equal the oracle's (IEEE divide)"""
    g = torch.Generator().manual_seed(77)
    rows, cols = 64, 1024
    ks = torch.arange(-140, 141, dtype=F32)
    scales = torch.tensor([0.125, 0.0390625, 3.0, 1.0 / 3.0, 0.1, 7.3e-5, 2.0 ** -90, 2.0 ** 90, 2.0 ** -110, 2.0 ** 110, 1e-3, 0.75, 5.0, 1.7, 2.0 ** -20, 9.5e4], dtype=F32)
    for bits, sym in ((4, True), (4, False), (8, True), (8, False), (6, False)):
        s = scales[torch.arange(rows) % scales.numel()].reshape(rows, 1).clone()
        x = torch.empty(rows, cols, dtype=F32)
        base = ((ks[torch.randint(0, ks.numel(), (rows, cols), generator=g)] + 0.5) * s)
        x.copy_(base)
        bits_view = x.view(torch.int32)
        bump = torch.randint(-2, 3, (rows, cols), generator=g, dtype=torch.int32)
        bits_view += bump  # exact ties and their 1-2 ulp neighbours
        x[:, -16:] = torch.randn(rows, 16, generator=g) * s * 3
        sp = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 3e38, -3e38, 1e-45, -1e-45, 1.17549435e-38], dtype=F32)
        x[:, : sp.numel()] = sp
        zp = torch.zeros(rows, 1, dtype=torch.int8) if sym else torch.randint(-(2 ** (bits - 1)), 2 ** (bits - 1), (rows, 1), generator=g, dtype=torch.int8)
        kw = dict(num_bits=bits, strategy="channel", group_size=None)
        q_ref = O.quantize(x, s, zp, dtype=torch.int8, **kw)
        q = cta.codec.quantize_tensor(x.to(dev), s.to(dev), zp.to(dev), dtype=torch.int8, **kw)
        assert eq(q.cpu(), q_ref), (bits, sym)
        assert eq(cta.codec.fake_quantize_tensor(x.to(dev), s.to(dev), zp.to(dev), **kw).cpu(), O.fake_quantize(x, s, zp, **kw)), (bits, sym)
        assert eq(cta.codec.quantize_and_pack(x.to(dev), s.to(dev), zp.to(dev), **kw).cpu(), O.pack_to_int32(q_ref, bits).contiguous()), (bits, sym)
        # group-wise with the same data: the W4 word kernel and the quads kernel index scales per group
        sg = s.expand(rows, cols // 128).contiguous()
        zg = zp.expand(rows, cols // 128).contiguous()
        kwg = dict(num_bits=bits, strategy="group", group_size=128)
        assert eq(cta.codec.quantize_and_pack(x.to(dev), sg.to(dev), zg.to(dev), **kwg).cpu(), O.pack_to_int32(O.quantize(x, sg, zg, dtype=torch.int8, **kwg), bits).contiguous()), (bits, sym)


def test_all_bf16_inputs_w4(cta, dev):
    """every bf16 bit pattern as an input, against a spread of scales incl. extreme exponents"""
    allx = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(BF16).reshape(64, 1024)
    g = torch.Generator().manual_seed(3)
    scales = torch.cat([
        (torch.rand(40, generator=g) * 2 + 1e-3), torch.tensor([1e-30, 1e30, 2.0 ** -70, 2.0 ** 70, 1e-38, 3e38, 0.0, float("inf")])
    ]).to(BF16)
    for i in range(0, scales.numel(), 8):
        s = scales[i:i + 8].reshape(1, 8).repeat(64, 1).contiguous()  # (64, 8): group 128
        z = torch.randint(-8, 8, (64, 8), generator=g).to(torch.int8)
        for zp in (None, z):
            ref = O.pack_to_int32(O.quantize(allx, s, zp, num_bits=4, strategy="group", group_size=128, dtype=torch.int8), 4)
            got = cta.codec.quantize_and_pack(allx.to(dev), s.to(dev), d(zp, dev), num_bits=4, strategy="group", group_size=128)
            assert eq(got.cpu(), ref.contiguous())


def test_all_fp16_inputs_w4_and_int8(cta, dev):
    """every fp16 bit pattern as an input (inf, NaN, subnormals, ties) through the packed-fp16 fast path of the fused W4 compress,
    the one-pass RTN entry and the int8 quantize kernels: scales inside, at both ends of and outside the proven range
    [2^-14, 2^15], with and without a (non-zero) zero point; rows that hold a non-finite weight take the exact form"""
    allx = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(F16).reshape(64, 1024)
    # a second copy whose rows are all finite except where the patterns say otherwise, shuffled so that lanes mix
    perm = torch.randperm(65536, generator=torch.Generator().manual_seed(5))
    shuffled = allx.reshape(-1)[perm].reshape(64, 1024).contiguous()
    g = torch.Generator().manual_seed(4)
    scales = torch.cat([
        (torch.rand(40, generator=g) * 2 + 1e-3), torch.tensor([2.0 ** -14, 2.0 ** -15, 2.0 ** 15, 1.5 * 2.0 ** 15, 6e-8, 65504.0, 0.0, float("inf")])
    ]).to(F16)
    for x in (allx, shuffled):
        for i in range(0, scales.numel(), 8):
            s = scales[i:i + 8].reshape(1, 8).repeat(64, 1).contiguous()  # (64, 8): group 128
            z = torch.randint(-8, 8, (64, 8), generator=g).to(torch.int8)
            for zp in (None, z, torch.zeros_like(z)):
                ref = O.pack_to_int32(O.quantize(x, s, zp, num_bits=4, strategy="group", group_size=128, dtype=torch.int8), 4)
                got = cta.codec.quantize_and_pack(x.to(dev), s.to(dev), d(zp, dev), num_bits=4, strategy="group", group_size=128)
                assert eq(got.cpu(), ref.contiguous()), (i, zp is None)
                z8 = None if zp is None else (zp * 9).to(torch.int8)
                for bits in (8, 6):
                    ref8 = O.quantize(x, s, z8, num_bits=bits, strategy="group", group_size=128, dtype=torch.int8)
                    got8 = cta.codec.quantize_tensor(x.to(dev), s.to(dev), d(z8, dev), num_bits=bits, strategy="group", group_size=128, dtype=torch.int8)
                    assert eq(got8.cpu(), ref8), (i, zp is None, bits)
    # one-pass round to nearest (observer + scale + quantize + pack) on fp16: finite data, the fast path end to end
    w = (torch.randn(96, 1024, generator=g) * 0.05).to(F16)
    for sym in (True, False):
        packed, scale, zp = cta.codec.rtn_quantize_and_pack(w.to(dev), group_size=128, symmetric=sym)
        s_ref, z_ref = O.calculate_qparams_minmax(w, num_bits=4, group_size=128, symmetric=sym)
        assert eq(scale.cpu(), s_ref) and eq(zp.cpu(), z_ref)
        assert eq(packed.cpu(), O.pack_to_int32(O.quantize(w, s_ref, z_ref, num_bits=4, strategy="group", group_size=128, dtype=torch.int8), 4).contiguous())


def test_all_fp16_inputs_float_typed_results(cta, dev):
    """fp16 reciprocal + Newton shortcut with its exact fix-up below 2^-13, where a float-typed result could see the difference (an
    underflowed zero's sign after `+ zero_point`): fake_quantize, quantize with a float result, float8 quantize — every fp16 input,
    scales that put the quotients inside, around and far below the fix-up threshold, zero points absent / all-zero / non-zero;
    bitwise against the oracle"""
    allx = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(F16).reshape(64, 1024)
    g = torch.Generator().manual_seed(8)
    scales = torch.cat([(torch.rand(8, generator=g) * 2 + 1e-3), torch.tensor([2.0 ** -14, 2.0 ** 15, 3.0e4, 1.0e4, 4097.0, 777.0, 0.37, 2.0 ** -15])]).to(F16)
    for i in range(0, scales.numel(), 8):
        s = scales[i:i + 8].reshape(1, 8).repeat(64, 1).contiguous()
        z = torch.randint(-8, 8, (64, 8), generator=g).to(torch.int8)
        for zp in (None, torch.zeros_like(z), z):
            kw = dict(num_bits=4, strategy="group", group_size=128)
            assert eq(cta.codec.fake_quantize_tensor(allx.to(dev), s.to(dev), d(zp, dev), **kw).cpu(), O.fake_quantize(allx, s, zp, **kw)), (i, zp is None)
            assert eq(cta.codec.quantize_tensor(allx.to(dev), s.to(dev), d(zp, dev), **kw).cpu(), O.quantize(allx, s, zp, **kw)), (i, zp is None)
            kw8 = dict(num_bits=8, strategy="group", group_size=128)
            assert eq(cta.codec.fake_quantize_tensor(allx.to(dev), s.to(dev), d(zp, dev), **kw8).cpu(), O.fake_quantize(allx, s, zp, **kw8)), (i, zp is None)
        for zf in (None, torch.zeros((64, 8), dtype=F8)):
            q = cta.codec.quantize_tensor(allx.to(dev), s.to(dev), d(zf, dev), num_bits=8, strategy="group", group_size=128, dtype=F8, qtype="float")
            r = O.quantize(allx, s, zf, num_bits=8, strategy="group", group_size=128, dtype=F8, qtype="float")
            nan = r != r
            assert torch.equal(q.cpu().view(torch.uint8)[~nan], r.view(torch.uint8)[~nan]) and bool((q.cpu() != q.cpu())[nan].all()), (i, zf is None)
    # channel-wise float8 round-to-nearest in one pass on fp16 (the scalar shortcut inside rtn_channel8)
    w = (torch.randn(64, 1024, generator=g) * 0.05).to(F16)
    w[3, :5] = torch.tensor([1e-7, -1e-7, 6e-8, -6e-8, 0.0]).to(F16)
    for qtype in ("float", "int"):
        qg, sg, zg = cta.codec.rtn_quantize_channel8(w.to(dev), qtype=qtype, symmetric=True)
        if qtype == "float":
            s_ref = O.calculate_qparams_float(w, kind="fp8")
            r = O.quantize(w, s_ref, torch.zeros_like(s_ref, dtype=F8), num_bits=8, strategy="channel", dtype=F8, qtype="float")
            assert eq(sg.cpu(), s_ref) and torch.equal(qg.cpu().view(torch.uint8), r.view(torch.uint8))
        else:
            s_ref, z_ref = O.calculate_qparams_minmax(w, num_bits=8, group_size=None, symmetric=True)
            assert eq(sg.cpu(), s_ref) and torch.equal(qg.cpu(), O.quantize(w, s_ref, z_ref, num_bits=8, strategy="channel", dtype=torch.int8))


@pytest.mark.parametrize("dtype", [BF16, F16, F32])
@pytest.mark.parametrize("cols", [5632, 11008, 96, 8 * 257])
def test_observer_on_rows_that_are_not_a_power_of_two(cta, dev, dtype, cols):
    """channel-wise min-max qparams on Llama-style widths (5632, 11008: the vectorised one-wave-per-row kernel), symmetric and not"""
    g = torch.Generator().manual_seed(cols)
    w = (torch.randn(33, cols, generator=g) * 0.3).to(dtype)
    w[1, 7] = 0.0
    for sym in (True, False):
        for bits in (8, 4):
            s, z = cta.codec.minmax_qparams(w.to(dev), num_bits=bits, group_size=None, symmetric=sym)
            s_ref, z_ref = O.calculate_qparams_minmax(w, num_bits=bits, group_size=None, symmetric=sym)
            assert eq(s.cpu(), s_ref) and eq(z.cpu(), z_ref), (sym, bits)


@pytest.mark.parametrize("dtype", [BF16, F16])
def test_asymmetric_decompress_full_range(cta, dev, dtype):
    """fused W4 decompress / int8 dequantize with an int8 zero point folded into the un-bias constant: every code x every zero
    point in [-128, 127] x a spread of scales, bit for bit against the oracle"""
    g = torch.Generator().manual_seed(6)
    rows, cols = 256, 1024  # 256 rows = one zero point each, 8 groups of 128 per row
    z = torch.arange(-128, 128, dtype=torch.int8).reshape(256, 1).repeat(1, 8).contiguous()
    s = (torch.rand((rows, 8), generator=g) * 0.3 + 1e-3).to(dtype)
    s[:, 0] = torch.tensor(2.0 ** -20).to(dtype)
    s[:, 1] = torch.tensor(300.0).to(dtype)
    q4 = torch.randint(-8, 8, (rows, cols), generator=g, dtype=torch.int8)
    packed = O.pack_to_int32(q4, 4).contiguous()
    got = cta.codec.unpack_and_dequantize(packed.to(dev), (rows, cols), s.to(dev), z.to(dev), num_bits=4)
    assert eq(got.cpu(), O.dequantize(q4, s, z))
    q8 = torch.randint(-128, 128, (rows, cols), generator=g, dtype=torch.int8)
    got8 = cta.codec.dequantize_tensor(q8.to(dev), s.to(dev), z.to(dev))
    assert eq(got8.cpu(), O.dequantize(q8, s, z))


@pytest.mark.parametrize("dtype", [BF16, F16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("shape,group", [((8192, 8192), 128), ((2048, 5632), 128), ((20, 512), 128), ((1001, 1024), 128), ((64, 256), 32), ((96, 384), 64), ((40, 160), 0)],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else f"g{v}")
def test_w4_packed_zero_points_ride_in_the_weights_launch(cta, dev, dtype, shape, group):
    """Round 6 (VERDICT r05 missing #4): the STORED form of an asymmetric scheme's zero points — pack_to_int32(zp, 4, packed_dim=0),
    compressors/pack_quantized/base.py:107-110 — is written by tail workgroups of the compress launch (`ct_quant_pack_w4_zp`) and read
    directly by the decompress launch, whose tail workgroups write the unpacked int8 form back (`ct_unpack_dequant_w4_zp`, base.py:147-153).
    Against the CPU oracle (pack_to_int32 is pinned to reference goldens there): packed weight words, packed zero-point words (rows that
    are not a multiple of 8: the padding nibbles are 0, not 8), decompressed weights, unpacked zero points; zero points at both ends of the
    int4 range in every nibble position.  group 0 = channel-wise."""
    from compressed_tensors_amd import _lib

    lib = _lib.load()
    rows, cols = shape
    g = group or cols
    G = cols // g
    gen = torch.Generator().manual_seed(rows + cols + g)
    big = rows * cols > 1 << 22
    w = torch.randn(shape, generator=gen).to(dtype) if not big else None
    wd = torch.randn(shape, generator=torch.Generator(device=dev).manual_seed(5), device=dev, dtype=torch.float32).to(dtype) if big else w.to(dev)
    sc, zp = cta.codec.minmax_qparams(wd, num_bits=4, group_size=group or None, symmetric=False)
    zp = zp.clone()
    zp[: min(rows, 16)] = torch.tensor([-8, 7, -1, 0, 3, -5, 6, -7, 7, -8, 1, 2, -3, 4, 5, -6], dtype=torch.int8, device=dev)[: min(rows, 16), None]  # every nibble position sees both ends
    stream = _lib.stream_of_device(dev)
    packed = torch.empty(rows, cols // 8, dtype=torch.int32, device=dev)
    zpp = torch.full(((rows * 4 + 31) // 32, G), -1, dtype=torch.int32, device=dev)
    rc = lib.ct_quant_pack_w4_zp(wd.data_ptr(), _lib.DT[dtype], sc.data_ptr(), zp.data_ptr(), rows, cols, g, packed.data_ptr(), zpp.data_ptr(), stream)
    assert rc == 0, _lib.last_error()
    kw = dict(num_bits=4, strategy="group" if group else "channel", group_size=group or None)
    assert torch.equal(packed, cta.codec.quantize_and_pack(wd, sc, zp, **kw))  # the two-launch composition (itself pinned below / elsewhere)
    assert torch.equal(zpp.cpu(), O.pack_to_int32(zp.cpu(), 4, packed_dim=0).contiguous())
    sl = slice(0, min(rows, 64))
    ref_c = O.pack_quantized_compress({"weight": wd[sl].cpu(), "weight_scale": sc[sl].cpu(), "weight_zero_point": zp[sl].cpu()}, symmetric=False, **kw)
    assert torch.equal(packed[sl].cpu(), ref_c["weight_packed"])
    if rows <= 64:
        assert torch.equal(zpp.cpu(), ref_c["weight_zero_point"])
    # decompress: the kernel takes the zero points from the stored words
    readable = g == 128 and cols % 512 == 0
    out = torch.empty(shape, dtype=dtype, device=dev)
    zu = torch.full((rows, G), 99, dtype=torch.int8, device=dev)
    rc = lib.ct_unpack_dequant_w4_zp(packed.data_ptr(), sc.data_ptr(), _lib.DT[dtype], zpp.data_ptr(), rows, cols, g, out.data_ptr(), zu.data_ptr(), stream)
    if not readable:
        assert rc == _lib.CT_ERR_UNSUPPORTED and "group == 128" in _lib.last_error()
        assert cta.codec.unpack_and_dequantize_with_zp(packed, shape, sc, zpp, num_bits=4) is None  # the codec-level form declines: the caller composes
    else:
        assert rc == 0, _lib.last_error()
        assert torch.equal(zu, zp)
        want = cta.codec.unpack_and_dequantize(packed, shape, sc, zp, num_bits=4)
        assert eq(out.cpu(), want.cpu())
        ref_d = O.pack_quantized_decompress(ref_c, num_bits=4, strategy=kw["strategy"], symmetric=False)
        assert eq(out[sl].cpu(), ref_d["weight"]) and torch.equal(zu[sl].cpu(), ref_d["weight_zero_point"])
        out2 = torch.empty_like(out)  # without the write-back (zp_out NULL): no tail workgroups, same weights
        assert lib.ct_unpack_dequant_w4_zp(packed.data_ptr(), sc.data_ptr(), _lib.DT[dtype], zpp.data_ptr(), rows, cols, g, out2.data_ptr(), None, stream) == 0
        assert eq(out2.cpu(), out.cpu())
    # the plug-in class takes these entries (one launch per direction) and still equals the oracle's state dicts
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, group_size=group or None, symmetric=False,
                                                                                       strategy="group" if group else "channel"))
    names = []
    real_call = cta.codec.call
    cta.codec.call = lambda name, *a: (names.append(name), real_call(name, *a))[1]
    try:
        comp = cta.PackedQuantizationCompressor.compress({"weight": wd, "weight_scale": sc, "weight_zero_point": zp}, scheme)
        n_c = len(names)
        dec = cta.PackedQuantizationCompressor.decompress(comp, scheme)
    finally:
        cta.codec.call = real_call
    assert names[:n_c] == ["ct_quant_pack_w4_zp"], names
    assert names[n_c:] == (["ct_unpack_dequant_w4_zp"] if readable else ["ct_unpack_int32_dim0", "ct_unpack_dequant"]), names
    assert torch.equal(comp["weight_packed"], packed) and torch.equal(comp["weight_zero_point"], zpp) and comp["weight_zero_point"].dtype == torch.int32
    assert torch.equal(dec["weight_zero_point"], zp) and dec["weight_zero_point"].dtype == torch.int8
    assert eq(dec["weight"].cpu(), cta.codec.unpack_and_dequantize(packed, shape, sc, zp, num_bits=4).cpu())


def test_w4_batch_mixes_items_with_and_without_stored_zero_points(cta, dev):
    """one `ct_quant_pack_batch` / `ct_unpack_dequant_batch` launch over a table that mixes symmetric items, asymmetric items whose zero
    points ride in the launch (zp_packed) and asymmetric items that keep the unpacked int8 form: every output equal to the single-tensor
    entries' (pinned against the oracle above)"""
    gen = torch.Generator(device=dev).manual_seed(3)
    shapes = [(2048, 2048), (256, 2048), (24, 512), (5632, 2048), (64, 1024), (2048, 5632), (8, 512)]
    items = []
    for i, (r, c) in enumerate(shapes):
        w = torch.randn(r, c, generator=gen, device=dev, dtype=torch.float32).to(BF16)
        sym = i % 3 == 0
        sc, zp = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=sym)
        items.append((w, sc, zp, sym, i % 3 == 1))
    kw = dict(num_bits=4, strategy="group", group_size=128)
    centries, want = [], []
    for w, sc, zp, sym, carry in items:
        r, c = w.shape
        pk = torch.empty(r, c // 8, dtype=torch.int32, device=dev)
        zpp = torch.empty((r * 4 + 31) // 32, c // 128, dtype=torch.int32, device=dev) if carry else None
        centries.append((w, sc, zp, pk, r, c, 128, zpp))
        want.append((cta.codec.quantize_and_pack(w, sc, zp, **kw), cta.codec.pack_to_int32(zp, 4, packed_dim=0)))
    cta.codec.W4Batch(centries, "compress", BF16).launch()
    for e, (wp, wz) in zip(centries, want):
        assert torch.equal(e[3], wp)
        if e[7] is not None:
            assert torch.equal(e[7], wz)
    dentries = []
    for (w, sc, zp, sym, carry), (wp, wz) in zip(items, want):
        r, c = w.shape
        out = torch.empty_like(w)
        if carry:  # stored form in, unpacked form written back
            dentries.append((wp, sc, torch.full_like(zp, 55), out, r, c, 128, wz))
        else:
            dentries.append((wp, sc, None if sym else zp, out, r, c, 128))
    cta.codec.W4Batch(dentries, "decompress", BF16).launch()
    for (w, sc, zp, sym, carry), e in zip(items, dentries):
        assert torch.equal(e[3], cta.codec.fake_quantize_tensor(w, sc, zp, **kw))  # value equality: fake_quantize may carry -0.0
        assert eq(e[3].cpu(), cta.codec.unpack_and_dequantize(e[0], w.shape, sc, None if sym else zp, num_bits=4).cpu())  # bit for bit: the single-tensor entry
        if carry:
            assert torch.equal(e[2], zp)


@pytest.mark.parametrize("N", [4096, 8192])
def test_w4a16_full_size(cta, dev, N):
    """BASELINE config 2 at full size: compress + decompress vs the oracle, and the round-trip
    property decompress(compress(W)) == fake_quantize(W) (reference test_pack_quant.py:160-183)"""
    torch.manual_seed(0)
    w = torch.randn(N, N, dtype=BF16)
    scale, zp = O.calculate_qparams_minmax(w, num_bits=4, group_size=128, symmetric=True)
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    sd = {"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}
    s_dev, z_dev = cta.codec.minmax_qparams(sd["weight"], num_bits=4, group_size=128, symmetric=True)
    assert eq(s_dev.cpu(), scale) and eq(z_dev.cpu(), zp)
    c = cta.PackedQuantizationCompressor.compress(sd, scheme)
    assert c["weight_packed"].shape == (N, N // 8) and "weight_zero_point" not in c
    q_ref = O.quantize(w, scale, zp, num_bits=4, strategy="group", group_size=128, dtype=torch.int8)
    assert torch.equal(c["weight_packed"].cpu(), O.pack_to_int32(q_ref, 4))
    dd = cta.PackedQuantizationCompressor.decompress(c, scheme)
    assert dd["weight"].dtype == BF16 and dd["weight"].shape == (N, N)
    # bitwise against the oracle's decompress; VALUE equality (torch.equal, as the reference's own
    # test does) against fake_quantize, whose -0.0 results come back as +0.0 through the int codes
    ref_d = O.pack_quantized_decompress({"weight_packed": c["weight_packed"].cpu(), "weight_scale": scale, "weight_shape": c["weight_shape"]},
                                        num_bits=4, strategy="group", symmetric=True)
    assert eq(dd["weight"].cpu(), ref_d["weight"])
    assert torch.equal(dd["weight"].cpu(), O.fake_quantize(w, scale, zp, num_bits=4, strategy="group", group_size=128))
    # idempotence: re-compressing the decompressed weight reproduces the same words
    c2 = cta.PackedQuantizationCompressor.compress({**sd, "weight": dd["weight"]}, scheme)
    assert torch.equal(c2["weight_packed"], c["weight_packed"])


@pytest.mark.parametrize("variant", ["fp32", "g_idx"])
def test_w4a16_full_size_other_layouts(cta, dev, variant):
    """the round-2 flat kernels at size, through the compressor class: fp32 weights and scales (4096x4096), and bf16 with activation
    ordering (weight_g_idx, 2048x8192) — packed words and the decompressed weight against the oracle, round trip == fake_quantize"""
    torch.manual_seed(3)
    if variant == "fp32":
        rows, cols, dt = 4096, 4096, F32
    else:
        rows, cols, dt = 2048, 8192, BF16
    w = torch.randn(rows, cols, dtype=torch.float32).to(dt)
    scale, zp = O.calculate_qparams_minmax(w, num_bits=4, group_size=128, symmetric=False)
    scale = scale.to(dt)
    g_idx = None
    if variant == "g_idx":
        g_idx = (torch.arange(cols, dtype=torch.int32) // 128)[torch.randperm(cols)].contiguous()
    kw = dict(num_bits=4, strategy="group", group_size=128)
    q_ref = O.quantize(w, scale, zp, dtype=torch.int8, g_idx=g_idx, **kw)
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=False, actorder="group" if g_idx is not None else None)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    sd = {"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}
    if g_idx is not None:
        sd["weight_g_idx"] = g_idx.to(dev)
    c = cta.PackedQuantizationCompressor.compress(sd, scheme)
    assert torch.equal(c["weight_packed"].cpu(), O.pack_to_int32(q_ref, 4))
    dd = cta.PackedQuantizationCompressor.decompress(c, scheme)
    ref = O.dequantize(q_ref, scale, zp, strategy="group", group_size=128, g_idx=g_idx)
    assert dd["weight"].dtype == dt and eq(dd["weight"].cpu(), ref)
    assert torch.equal(dd["weight"].cpu(), O.fake_quantize(w, scale, zp, g_idx=g_idx, **kw))


@pytest.mark.parametrize("shape", [(2048, 5632), (8192, 4096), (4096, 8192 + 128)], ids=["u4", "u8_exact_round", "u8"])
@pytest.mark.parametrize("kind", ["w4_sym", "w4_asym", "int8", "w8_packed"])
def test_decompress_side_one_residency_round(cta, dev, shape, kind):
    """tensors that fit one residency round are decompressed with 4 or 8 units per lane (`decomp_unroll`, round 3): every kernel
    that takes the unroll — W4 symmetric, W4 asymmetric (the row-leader form), int8 dequantize, 8-bit packed — against the oracle"""
    rows, cols = shape
    g = torch.Generator().manual_seed(rows + cols)
    w = torch.randn(rows, cols, generator=g).to(BF16)
    if kind.startswith("w4") or kind == "w8_packed":
        bits = 4 if kind.startswith("w4") else 8
        sym = kind != "w4_asym"
        scale, zp = O.calculate_qparams_minmax(w, num_bits=bits, group_size=128, symmetric=sym)
        kw = dict(num_bits=bits, strategy="group", group_size=128)
        q = O.quantize(w, scale, zp, dtype=torch.int8, **kw)
        packed = O.pack_to_int32(q, bits).contiguous()
        got = cta.codec.unpack_and_dequantize(packed.to(dev), (rows, cols), scale.to(dev), None if sym else zp.to(dev), **kw)
        assert eq(got.cpu(), O.dequantize(q, scale, None if sym else zp, strategy="group", group_size=128))
    else:
        scale, zp = O.calculate_qparams_minmax(w, num_bits=8, group_size=None, symmetric=True)
        q = O.quantize(w, scale, zp, num_bits=8, strategy="channel", dtype=torch.int8)
        got = cta.codec.dequantize_tensor(q.to(dev), scale.to(dev), None)
        assert eq(got.cpu(), O.dequantize(q, scale, None))


@pytest.mark.parametrize("case", ["sym", "asym", "asym_fp16", "scale_view_off_by_2_bytes", "zp_view_off_by_1_byte", "group_64", "units_not_a_multiple_of_64"])
def test_w4_decompress_scalar_scale_loads_and_their_fallbacks(cta, dev, case):
    """round 5: a W4 tensor of many residency rounds (> 33.5 M elements) with groups of 128 fetches a wave's four scales / zero points by
    scalar loads (s_load_dwordx2 / s_load_dword); anything else — another group size, units % 64 != 0, a scale table that is not 8-byte or
    a zero-point table that is not 4-byte aligned — keeps the vector loads.  Every case must give the bits of the SAME tensor decompressed
    in row blocks small enough to take the one-round kernels (vector loads), and its first rows the oracle's bits"""
    rows, cols, gs, dt = 4160, 8192, 128, BF16
    if case == "group_64":
        gs = 64
    if case == "units_not_a_multiple_of_64":
        rows, cols = 4101, 8192 + 128
    if case == "asym_fp16":
        dt = F16
    sym = case in ("sym", "group_64")
    assert rows * cols > 8 * 256 * 8 * 256 * 8
    g = torch.Generator(device=dev).manual_seed(len(case))
    packed = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, cols // 8), dtype=torch.int32, device=dev, generator=g)
    groups = cols // gs
    sbuf = (torch.rand(rows * groups + 8, device=dev, generator=g) * 0.3 + 1e-3).to(dt)
    zbuf = torch.randint(-8, 8, (rows * groups + 8,), dtype=torch.int8, device=dev, generator=g)
    so = 1 if case == "scale_view_off_by_2_bytes" else 0
    zo = 1 if case == "zp_view_off_by_1_byte" else 0
    scale = sbuf[so:so + rows * groups].view(rows, groups)
    zp = None if sym else zbuf[zo:zo + rows * groups].view(rows, groups)
    if so:
        assert scale.data_ptr() % 8 != 0
    kw = dict(num_bits=4, strategy="group", group_size=gs)
    got = cta.codec.unpack_and_dequantize(packed, (rows, cols), scale, zp, **kw)
    step = 1024  # 8.4 M elements per block: one residency round, the vector-load kernels
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        blk = cta.codec.unpack_and_dequantize(packed[r0:r1].contiguous(), (r1 - r0, cols), scale[r0:r1].contiguous(), None if zp is None else zp[r0:r1].contiguous(), **kw)
        assert eq(got[r0:r1], blk), (case, r0)
    q = O.unpack_from_int32(packed[:64].cpu(), 4, (64, cols))
    assert eq(got[:64].cpu(), O.dequantize(q, scale[:64].cpu(), None if zp is None else zp[:64].cpu(), strategy="group", group_size=gs))
    # the batched entry takes the same per-item decision (two items, the second with a table that keeps the vector loads)
    half = rows // 2 // 64 * 64
    items = [(packed[:half].contiguous(), (half, cols), scale[:half].contiguous(), None if zp is None else zp[:half].contiguous()),
             (packed[half:].contiguous(), (rows - half, cols), sbuf[1:1 + (rows - half) * groups].view(rows - half, groups), None if zp is None else zp[half:].contiguous())]
    outs = cta.codec.unpack_and_dequantize_many(items, **kw)
    assert eq(outs[0], got[:half])
    ref1 = cta.codec.unpack_and_dequantize(items[1][0], items[1][1], items[1][2].contiguous(), items[1][3], **kw)
    assert eq(outs[1], ref1)


def test_int8_per_tensor_full_size(cta, dev):
    """BASELINE config 1 on the GPU path: int8 per-tensor symmetric, 4096x4096 bf16"""
    torch.manual_seed(0)
    w = torch.randn(4096, 4096, dtype=BF16)
    scale, zp = O.calculate_qparams_minmax(w.reshape(1, -1), num_bits=8, symmetric=True)
    scale, zp = scale.reshape(1), zp.reshape(1)
    args = cta.QuantizationArgs(num_bits=8, strategy="tensor", symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args, input_activations=cta.QuantizationArgs(num_bits=8))
    comp = cta.BaseCompressor.get_value_from_registry("int-quantized")
    c = comp.compress({"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}, scheme)
    q_ref = O.quantize(w, scale, zp, num_bits=8, strategy="tensor", dtype=torch.int8)
    assert c["weight"].dtype == torch.int8 and torch.equal(c["weight"].cpu(), q_ref)
    dd = comp.decompress(c, scheme)
    assert eq(dd["weight"].cpu(), O.dequantize(q_ref, scale, None))
    assert torch.equal(dd["weight"].cpu(), O.fake_quantize(w, scale, zp, num_bits=8, strategy="tensor"))


# ----------------------------------------------------------------------------- sparse
@pytest.mark.parametrize("case", [c for c in cases("sparse") if c["kind"] == "bitmask"], ids=lambda c: c["key"])
def test_bitmask_primitives_golden(golden, cta, dev, case):
    t = golden.case("sparse", case["key"])
    assert eq(cta.codec.pack_bitmasks(t["mask"].bool().to(dev)).cpu(), t["packed"])
    assert torch.equal(cta.codec.unpack_bitmasks(t["packed"].to(dev), case["shape"]).cpu(), t["mask"].bool())


@pytest.mark.parametrize("dtype", [BF16, F16, F32, torch.int8])
@pytest.mark.parametrize("shape,p", [((1, 1), 0.5), ((3, 10), 0.5), ((16, 64), 0.5), ((7, 129), 0.3), ((64, 4096), 0.5),
                                     ((5, 2048), 0.0), ((5, 2048), 1.0), ((9, 6000), 0.9),
                                     # 16-bit fast path (cols % 32 == 0): one tile, exactly one tile, several tiles per row
                                     ((33, 32), 0.5), ((6, 8192), 0.5), ((3, 8192), 0.0), ((3, 8192), 1.0), ((5, 8192 + 32), 0.4),
                                     ((3, 3 * 8192 + 4096), 0.5), ((2, 2 * 8192), 0.0), ((257, 96), 0.7)])
def test_bitmask_codec_vs_oracle(cta, dev, dtype, shape, p):
    g = torch.Generator().manual_seed(shape[1])
    x = torch.randn(shape, generator=g)
    x = x.masked_fill(torch.rand(shape, generator=g) < p, 0)
    x = (x * 8).to(dtype) if dtype is torch.int8 else x.to(dtype)
    if dtype.is_floating_point and x.numel() > 2:
        x.view(-1)[0] = -0.0
        x.view(-1)[1] = float("nan")
    rv, rb, ro = O.bitmask_compress(x)
    values, bitmask, row_offsets = cta.codec.bitmask_compress(x.to(dev))
    assert eq(values.cpu(), rv) and torch.equal(bitmask.cpu(), rb) and torch.equal(row_offsets.cpu(), ro)
    ref = O.bitmask_decompress(rv, rb, shape)
    assert eq(cta.codec.bitmask_decompress(values, bitmask, shape, row_offsets).cpu(), ref)
    # a checkpoint without row_offsets: rebuilt from the bitmask
    assert eq(cta.codec.bitmask_decompress(values, bitmask, shape).cpu(), ref)


@pytest.mark.parametrize("shape", [(300, 8192), (7, 5 * 8192 + 24), (2000, 64), (64, 4096), (1100, 8192 * 2), (1, 8), (3, 8), (5, 40),
                                   (4099, 264), (2, 1 << 20)])
def test_bitmask_fused_vs_two_pass(cta, dev, shape):
    """fused flat form (span / block counts + scatter) and the count / scan / host read / scatter form
    both match the oracle"""
    g = torch.Generator().manual_seed(shape[0])
    x = torch.randn(shape, generator=g).masked_fill(torch.rand(shape, generator=g) < 0.6, 0).to(BF16)
    rv, rb, ro = O.bitmask_compress(x)
    for two_pass in (False, True):
        values, bitmask, row_offsets = cta.codec.bitmask_compress(x.to(dev), two_pass=two_pass)
        assert eq(values.cpu(), rv) and torch.equal(bitmask.cpu(), rb) and torch.equal(row_offsets.cpu(), ro)


def test_bitmask_fused_capacity_guard(cta, dev):
    """a too-small value buffer is never overrun and the needed size is still reported"""
    from compressed_tensors_amd import _lib
    lib = _lib.load()
    x = torch.randn(64, 8192).masked_fill(torch.rand(64, 8192) < 0.5, 0).to(BF16).to(dev)
    rv, _, _ = O.bitmask_compress(x.cpu())
    cap = 1000
    buf = torch.full((cap + 4096,), 0x7fc0, dtype=torch.int16, device=dev)
    bm = torch.empty(64, 1024, dtype=torch.uint8, device=dev)
    ro = torch.empty(64, dtype=torch.int64, device=dev)
    nbytes = int(lib.ct_bitmask_compress_workspace_bytes(64, 8192))
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.int64, device=dev)
    _lib.call("ct_bitmask_compress", x.data_ptr(), _lib.BF16, 64, 8192, buf.data_ptr(), cap, bm.data_ptr(), ro.data_ptr(),
              ws[-1:].data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    assert int(ws[-1].item()) == rv.numel()
    assert torch.equal(buf[:cap].cpu(), rv.view(torch.int16)[:cap])
    assert bool((buf[cap:] == 0x7fc0).all())


def test_bitmask_full_size(cta, dev):
    """BASELINE config 3: 50 % unstructured, 8192x8192 bf16"""
    N = 8192
    torch.manual_seed(0)
    w = torch.randn(N, N, dtype=BF16)
    w = w.masked_fill(torch.rand(N, N, generator=torch.Generator().manual_seed(1)) < 0.5, 0)
    rv, rb, ro = O.bitmask_compress(w)
    t = cta.compressors.sparse.BitmaskTensor.from_dense(w.to(dev))
    assert eq(t.compressed.cpu(), rv) and torch.equal(t.bitmask.cpu(), rb) and torch.equal(t.row_offsets.cpu(), ro)
    assert eq(t.decompress().cpu(), w)  # no -0.0 in randn*mask: exact round trip
    # the same three tensors restated with eager torch ops on the device, independent of the oracle (the definition of the format:
    # values = x[x != 0] in row-major order, bitmask = little-endian packbits of x != 0 — `pack_bitmasks`, utils/helpers.py:306-343 —,
    # row_offsets = exclusive cumsum of the per-row counts)
    wd = w.to(dev)
    mask = wd != 0
    assert torch.equal(t.compressed, wd[mask])
    weights = (1 << torch.arange(8, device=dev, dtype=torch.int32))
    assert torch.equal(t.bitmask, (mask.view(N, N // 8, 8).to(torch.int32) * weights).sum(-1).to(torch.uint8))
    counts = mask.sum(-1)
    assert torch.equal(t.row_offsets, torch.cumsum(counts, 0) - counts)
    # 2:4 codec at the same size: values = the two largest magnitudes of every four, in order; bitmask as above; decompress restores them
    m24 = cta.codec.sparse24_mask(wd.abs().to(BF16) + 0 * wd)  # top-2 of |x| (ties to the lower index, as torch.topk)
    pruned = wd * m24.to(wd.dtype)
    v24, b24 = cta.codec.sparse24_bitmask_compress(pruned)
    assert torch.equal(b24, (m24.view(N, N // 8, 8).to(torch.int32) * weights).sum(-1).to(torch.uint8)) and bool((m24.view(-1, 4).sum(-1) == 2).all())
    assert torch.equal(v24.reshape(-1), pruned[m24.bool()]) and torch.equal(cta.codec.sparse24_bitmask_decompress(v24, b24, (N, N)), pruned)


@pytest.mark.parametrize("shape", [(3, 4), (5, 12), (1, 4), (7, 20), (2, 2, 3)])
def test_sparse24_mask_accepts_any_multiple_of_four(cta, dev, shape):
    """mask_creator (utils/semi_structured_conversions.py:301-330) takes numel % 4 == 0, not only % 8"""
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(BF16)
    assert torch.equal(cta.codec.sparse24_mask(x.to(dev)).cpu(), O.sparse24_mask(x))
    with pytest.raises(ValueError):
        cta.codec.sparse24_mask(torch.zeros(3, 2, dtype=BF16, device=dev))


def test_marlin24_batch_with_a_violation_leaves_every_module_untouched(cta, dev):
    """Marlin24Compressor.compress_modules validates the whole batch before it replaces anything, as upstream validates
    before mutating — a batch with one non-2:4 weight raises the ValueError (naming the module) and leaves all modules dense"""
    torch.manual_seed(2)
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    mods = []
    for k in range(3):
        w = torch.randn(64, 256, dtype=BF16, device=dev)
        if k != 1:
            w = w * cta.codec.sparse24_mask(w).to(w.dtype)  # module 1 stays dense: violates 2:4
        lin = torch.nn.Linear(256, 64, bias=False, device="meta")
        lin.weight = torch.nn.Parameter(w, requires_grad=False)
        sc, zp = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
        lin.weight_scale = torch.nn.Parameter(sc, requires_grad=False)
        lin.weight_zero_point = torch.nn.Parameter(zp, requires_grad=False)
        lin.quantization_scheme = scheme
        mods.append(lin)
    before = [(m.weight.data_ptr(), sorted(m._parameters)) for m in mods]
    with pytest.raises(ValueError, match="layers.1.proj"):
        cta.Marlin24Compressor.compress_modules(mods, names=["layers.0.proj", "layers.1.proj", "layers.2.proj"])
    assert [(m.weight.data_ptr(), sorted(m._parameters)) for m in mods] == before
    assert all(getattr(m, "quantization_status", None) is None for m in mods)
    good = [mods[0], mods[2]]
    cta.Marlin24Compressor.compress_modules(good)
    assert all(sorted(k for k, v in m._parameters.items() if v is not None) == ["meta", "scale_packed", "weight_packed"] for m in good)


@pytest.mark.parametrize("dtype", [BF16, F16, torch.int8])
def test_sparse24_vs_oracle(cta, dev, dtype):
    g = torch.Generator().manual_seed(6)
    x = torch.randn((64, 256), generator=g)
    x = (x * 20).to(dtype) if dtype is torch.int8 else x.to(dtype)
    x[0, :8] = 0  # all-zero quads still get two bits
    x[1, :4] = x[1, 0]  # ties
    m = cta.codec.sparse24_mask(x.to(dev))
    assert torch.equal(m.cpu(), O.sparse24_mask(x))
    pruned = x * O.sparse24_mask(x).to(x.dtype)
    rv, rb = O.sparse24_bitmask_compress(pruned)
    values, bitmask = cta.codec.sparse24_bitmask_compress(pruned.to(dev))
    assert eq(values.cpu(), rv) and torch.equal(bitmask.cpu(), rb)
    out = cta.codec.sparse24_bitmask_decompress(values, bitmask, pruned.shape)
    assert eq(out.cpu(), O.sparse24_bitmask_decompress(rv, rb, pruned.shape))


@pytest.mark.parametrize("dtype", [torch.int8, torch.float8_e4m3fn, BF16, F32], ids=["int8", "fp8", "bf16", "fp32"])
@pytest.mark.parametrize("cols", [64, 1040, 2080, 8192, 11008, 16384 + 64])
def test_sparse24_decompress_regular_and_irregular_rows(cta, dev, dtype, cols):
    """the general bitmask decompress expands 2:4-regular rows locally (no prefix) and every other row through the prefix path; a
    tensor whose rows all keep cols / 2 elements but only some of which are 2:4-regular must come back exactly (oracle: the dense
    scatter), for 8-, 16- and 32-bit payloads"""
    g = torch.Generator().manual_seed(cols)
    rows = 37
    base = torch.randn(rows, cols, generator=g)
    mask = torch.zeros(rows, cols, dtype=torch.bool)
    quads = mask.view(rows, cols // 4, 4)
    for r in range(rows):
        if r % 3 == 1:  # irregular row: the same number of kept elements, placed anywhere
            idx = torch.randperm(cols, generator=g)[: cols // 2]
            mask[r, idx] = True
        else:          # regular: two of every four
            sel = torch.rand(cols // 4, 4, generator=g).argsort(dim=-1)[:, :2]
            quads[r].scatter_(1, sel, True)
    if dtype in (torch.int8,):
        dense = (base * 40).to(torch.int8)
        dense[dense == 0] = 1
    elif dtype is torch.float8_e4m3fn:
        dense = (base.to(BF16).to(dtype).view(torch.int8) | 1).view(dtype)  # no zero payloads
    else:
        dense = (base + base.sign() * 0.5).to(dtype)
    dense_bits = dense.view(torch.int8) if dense.element_size() == 1 else dense
    pruned = torch.where(mask, dense_bits, torch.zeros_like(dense_bits))
    values = pruned[mask].reshape(rows, cols // 2)
    weights = (1 << torch.arange(8, dtype=torch.int32))
    bitmask = (mask.view(rows, cols // 8, 8).to(torch.int32) * weights).sum(-1).to(torch.uint8)
    out = cta.codec.sparse24_bitmask_decompress(values.to(dev), bitmask.to(dev), (rows, cols))
    assert torch.equal(out.cpu(), pruned)
    ref = O.sparse24_bitmask_decompress(values, bitmask, (rows, cols))
    assert torch.equal(out.cpu().view(torch.uint8), ref.view(torch.uint8))


@pytest.mark.parametrize("case", [c for c in cases("sparse") if c["kind"] == "cutlass24"], ids=lambda c: c["key"])
def test_cutlass24_golden(golden, cta, dev, case):
    t = golden.case("sparse", case["key"])
    sparse, meta = cta.codec.cutlass24_from_dense(t["dense"].to(dev))
    assert eq(sparse.cpu(), t["sparse"]) and eq(meta.cpu(), t["meta"])
    assert eq(cta.codec.cutlass24_to_dense(t["sparse"].to(dev), t["meta"].to(dev)).cpu(), t["dense_rt"])


@pytest.mark.parametrize("bits,strategy,gs", [(4, "group", 128), (4, "channel", None), (8, "channel", None)])
def test_marlin24_vs_oracle(cta, dev, bits, strategy, gs):
    g = torch.Generator().manual_seed(bits)
    out_f, in_f = 128, 512
    w = torch.randn((out_f, in_f), generator=g).to(BF16)
    w = w * O.sparse24_mask(w).to(w.dtype)
    scale, zp = O.calculate_qparams_minmax(w.to(F16), num_bits=bits, group_size=gs, symmetric=True)
    ref = O.marlin24_compress(w, scale, zp, num_bits=bits, strategy=strategy, group_size=gs)
    args = cta.QuantizationArgs(num_bits=bits, strategy=strategy, group_size=gs, symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    got = cta.Marlin24Compressor.compress({"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}, scheme)
    assert sorted(got.keys()) == ["meta", "scale_packed", "weight_packed"]
    for k in ref:
        assert got[k].shape == ref[k].shape, k
        assert eq(got[k].cpu().contiguous(), ref[k].contiguous()), k
    # the permutation tables themselves
    perm, sp, sps = cta.utils.get_permutations_24(bits)
    assert torch.equal(perm, O.marlin24_perm(bits).long())


def _marlin24_case(cta, dev, out_f, in_f, bits, strategy, gs, wdt=BF16, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn((out_f, in_f), generator=g).to(wdt)
    w = w * O.sparse24_mask(w).to(w.dtype)
    scale, zp = O.calculate_qparams_minmax(w.to(F16), num_bits=bits, group_size=gs, symmetric=True)
    ref = O.marlin24_compress(w, scale, zp, num_bits=bits, strategy=strategy, group_size=gs)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=bits, strategy=strategy, group_size=gs, symmetric=True))
    got = cta.Marlin24Compressor.compress({"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}, scheme)
    assert sorted(got.keys()) == ["meta", "scale_packed", "weight_packed"]
    for k in ref:
        assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype, (k, got[k].shape, ref[k].shape)
        assert eq(got[k].cpu().contiguous(), ref[k].contiguous()), k
    return got


def test_marlin24_full_size(cta, dev):
    """BASELINE config 4 at its stated size: 2:4 + int4 g128 marlin-24 packing of 8192x8192 bf16 — packed words,
    permuted fp16 scales and reordered 2:4 metadata bit-exact with the oracle (64-bit offsets, all 1 048 576 threads
    of marlin24_fused_w4_kernel).  Spec: utils/semi_structured_conversions.py:66-197, utils/permutations_24.py:20-53."""
    got = _marlin24_case(cta, dev, 8192, 8192, 4, "group", 128)
    assert got["weight_packed"].shape == (8192 // 2 // 16, 8192 * 16 // 8) and got["meta"].shape == (8192 // 16 // 2, 8192 * 2)
    assert got["scale_packed"].shape == (64, 8192) and got["scale_packed"].dtype == F16


@pytest.mark.parametrize("out_f,in_f,bits,strategy,gs", [
    (512, 2048, 4, "group", 128), (2048, 512, 4, "group", 128),   # non-square, fused one-launch path (in % 256 == 0)
    (192, 1024, 4, "channel", None), (1024, 320, 4, "group", 32),  # in % 256 != 0: front end + packing kernel
    (256, 2048, 8, "group", 128), (2048, 256, 8, "channel", None),  # int8 codes
    (128, 256, 4, "group", 128),   # group_size == in/2 == size_k of the compressed weight: the single-column scale permutation
    (128, 512, 4, "group", 256),   # same boundary, one level up
])
@pytest.mark.parametrize("wdt", [BF16, F16])
def test_marlin24_non_square(cta, dev, out_f, in_f, bits, strategy, gs, wdt):
    """`scale_packed` is (groups, out_features) and the group permutation applies iff group_size < in_features / 2
    (size_k of the compressed, transposed weight) — the layout vLLM's marlin-24 loader expects; see DESIGN.md 5.6"""
    got = _marlin24_case(cta, dev, out_f, in_f, bits, strategy, gs, wdt=wdt, seed=out_f + in_f)
    groups = in_f // gs if strategy == "group" else 1
    assert got["scale_packed"].shape == (groups, out_f)
    assert got["weight_packed"].shape == (in_f // 2 // 16, out_f * 16 // (32 // bits))
    assert got["meta"].shape == (in_f // 16 // 2, out_f * 2)


@pytest.mark.parametrize("out_f,in_f,strategy,gs", [(8192, 8192, "group", 128), (512, 2048, "group", 128), (256, 1024, "channel", None)])
def test_marlin24_decodes_through_the_reference_primitives(cta, dev, out_f, in_f, strategy, gs):
    """encode -> decode property at BASELINE config 4's size, independent of the oracle's restatement of the (removed) compressor
    class: the three stored tensors are taken apart with the primitives that DO survive in the reference and are golden-pinned
    here — the inverse of `get_permutations_24`'s tile permutation (utils/permutations_24.py:20-53) and
    `sparse_semi_structured_to_dense_cutlass` (utils/semi_structured_conversions.py:204-298) — and must give back the fp16
    fake-quantized weight: codes x scales == fake_quantize(W.to(fp16))."""
    g = torch.Generator().manual_seed(out_f + in_f)
    w = torch.randn((out_f, in_f), generator=g).to(BF16)
    w = w * O.sparse24_mask(w).to(w.dtype)
    scale, zp = O.calculate_qparams_minmax(w.to(F16), num_bits=4, group_size=gs, symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, strategy=strategy, group_size=gs, symmetric=True))
    got = cta.Marlin24Compressor.compress({"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}, scheme)
    k2, n = in_f // 2, out_f
    perm, sperm, sperm_single = cta.utils.get_permutations_24(4)
    perm = perm.to(dev)
    # weight_packed -> unsigned nibbles -> undo the tile permutation -> untile -> (n, k/2) signed kept codes
    wp = got["weight_packed"]
    assert wp.shape == (k2 // 16, n * 16 // 8)
    nib = torch.stack([(wp >> (4 * i)) & 0xF for i in range(8)], dim=-1).reshape(k2 // 16, n * 16)
    t0 = torch.empty_like(nib).reshape(-1, 1024)
    t0[:, perm] = nib.reshape(-1, 1024)
    codes_t = t0.reshape(k2 // 16, n // 16, 16, 16).permute(0, 2, 1, 3).reshape(k2, n)
    comp = (codes_t.t().contiguous() - 8).to(F16)  # (n, k/2) kept codes as the fp16 values the reference's 2:4 routines take
    # meta: undo the final view, then the reference's CUTLASS decode
    meta = got["meta"].reshape(-1).reshape(n, in_f // 16)
    dense_codes = cta.codec.cutlass24_to_dense(comp, meta)
    assert dense_codes.shape == (n, in_f)
    assert bool(((dense_codes != 0).view(-1, 4).sum(-1) <= 2).all())
    # scale_packed -> undo the scale permutation and the transpose
    sp = got["scale_packed"]
    groups = in_f // gs if strategy == "group" else 1
    assert sp.shape == (groups, n) and sp.dtype == F16
    tbl = torch.tensor(sperm if (strategy == "group" and gs < k2) else sperm_single, device=dev)
    st = torch.empty_like(sp).reshape(-1, 64)
    st[:, tbl] = sp.reshape(-1, 64)
    scale_back = st.reshape(groups, n).t().contiguous()
    assert torch.equal(scale_back.cpu(), scale.to(F16))
    deq = (dense_codes.reshape(n, groups, -1) * scale_back.unsqueeze(-1)).reshape(n, in_f)
    want = O.fake_quantize(w.to(F16), scale.to(F16), zp, num_bits=4, strategy=strategy, group_size=gs)
    assert torch.equal(deq.cpu(), want)  # value equality (an all-zero quad comes back as +0.0)


def test_marlin24_rejects_dense_weight(cta, dev):
    """a weight that is not 2:4 must be refused (fused front end: device flag, one host read)"""
    w = torch.randn(64, 256).to(BF16)
    scale, zp = O.calculate_qparams_minmax(w.to(F16), num_bits=4, group_size=128, symmetric=True)
    args = cta.QuantizationArgs(num_bits=4, strategy="group", group_size=128, symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    with pytest.raises(ValueError, match="2:4 sparsity structure"):
        cta.Marlin24Compressor.compress({"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}, scheme)


@pytest.mark.parametrize("native", [True, False], ids=["cpp-host", "python-host"])
def test_from_dense_values_own_exactly_nnz_elements_by_default(cta, dev, native):
    """`tensor[mask]` returns nnz elements (restated S1 over utils/helpers.py:306-343) — so does
    `BitmaskTensor.from_dense` by default, at every density: the storage behind `compressed` is nnz x itemsize bytes (+ the allocator's
    rounding), not the dense-sized buffer the kernel wrote into.  `exact=False` is the opt-in view (no copy, dense-sized storage)."""
    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd.compressors.sparse.sparse_bitmask import BitmaskTensor

    hp = ctlib.hostpath()
    assert hp is not None
    g = torch.Generator(device=dev).manual_seed(11)
    ctlib._HOSTPATH[0] = hp if native else None
    try:
        for dtype in (BF16, F32):
            for density in (0.5, 0.9, 0.375, 0.05, 0.0, 1.0):
                w = torch.randn(1024, 2048, device=dev, generator=g).to(dtype)
                w = w * (torch.rand(1024, 2048, device=dev, generator=g) < density) if density < 1.0 else w.abs() + 1
                nnz = int((w != 0).sum())
                bt = BitmaskTensor.from_dense(w)
                assert bt.compressed.numel() == nnz and torch.equal(bt.compressed, w[w != 0])
                assert bt.compressed.untyped_storage().nbytes() <= nnz * w.element_size() + (2 << 20), (dtype, density)
                assert torch.equal(bt.decompress(), w)
                view = BitmaskTensor.from_dense(w, exact=False)
                assert torch.equal(view.compressed, bt.compressed) and torch.equal(view.bitmask, bt.bitmask) and torch.equal(view.row_offsets, bt.row_offsets)
                assert view.compressed.untyped_storage().nbytes() == w.numel() * w.element_size()
    finally:
        ctlib._HOSTPATH[0] = hp


@pytest.mark.parametrize("dtype", [torch.int8, torch.float8_e4m3fn, torch.uint8], ids=["int8", "fp8", "uint8"])
def test_sparse_bitmask_8bit_payloads_take_the_resident_kernel(cta, dev, dtype):
    """Round 6 (VERDICT r05 missing #5): 8-bit payloads — FP8 / int8 weights, gathered as raw bytes (restated S1 over utils/helpers.py:306-343:
    "FP8 viewed as int8 for the gather") — ride the one-pass resident kernel's row form when a row is whole 16-byte units (cols % 16 == 0): a unit is
    16 elements = two bitmask bytes, the counts are in bytes, values leave at byte granularity.  Against the CPU oracle and eager torch on the
    device: one workgroup, partial last tiles, rows that straddle tiles, all-zero / dense tensors, every density, 8192 x 8192 (two residency
    rounds); cols % 16 != 0 keeps the count / scan / scatter form; the batched entry takes 8-bit tables too.  The decompress side takes the
    byte-granular LDS-window kernel (single-tile rows, rows of several 16384-column tiles, flat tiles over rows shorter than 8192 columns, value
    runs that start at every offset inside a 16-byte vector, a run that ends at the very end of the buffer)."""
    g = torch.Generator(device=dev).manual_seed(8)

    def make(r, c, dens):
        raw = torch.randint(-127, 128, (r, c), device=dev, generator=g, dtype=torch.int16).to(torch.int8)
        raw = raw * (torch.rand(r, c, device=dev, generator=g) < dens) if dens < 1.0 else raw.abs().clamp(min=1)
        if dtype is torch.float8_e4m3fn:
            raw = raw.masked_fill(raw == -128, 0)  # (0x80 = -0.0 in fp8 would be a kept byte: bytes are compared as bytes, as before)
        return raw.view(dtype) if dtype is not torch.int8 else raw

    cases = [(1, 16, 0.5), (3, 48, 0.3), (64, 256, 0.0), (257, 1008, 0.7), (2048, 2048, 0.5), (100, 64, 1.0), (33, 4096, 0.02), (5632, 2048, 0.5), (8192, 8192, 0.5),
             (17, 40, 0.5), (64, 1000, 0.5),  # these two: cols % 16 != 0 -> count / scan / scatter
             (3, 16384, 0.5), (5, 32768 + 64, 0.4), (2, 65536, 0.95), (7, 16384 + 4096, 0.0),  # rows of one whole tile / several tiles of the byte-window decompress
             (33, 4096, 0.5), (9, 4096 + 16, 0.6), (11, 8192 + 48, 0.5), (5, 12288, 0.3), (4, 16384 - 16, 0.8),  # 1 / 2 / 4 units per lane, partial tiles
             (37, 2080, 0.5), (1000, 96, 0.4), (5, 8160, 0.6), (129, 32, 0.5)]  # flat tiles (cols < 8192, cols % 32 == 0): rows that straddle tiles, a partial last tile
    ws = [make(*c) for c in cases]
    for w in ws:
        v, bm, ro = cta.codec.bitmask_compress(w)
        b = w.view(torch.uint8)
        keep = b != 0
        assert v.dtype == w.dtype and torch.equal(v.view(torch.uint8), b[keep]), tuple(w.shape)
        cnt = keep.sum(-1)
        assert torch.equal(ro, torch.cumsum(cnt, 0) - cnt)
        weights = (1 << torch.arange(8, device=dev, dtype=torch.int32))
        pad = (-w.shape[1]) % 8
        kp = torch.nn.functional.pad(keep, (0, pad)) if pad else keep
        assert torch.equal(bm, (kp.view(w.shape[0], -1, 8).to(torch.int32) * weights).sum(-1).to(torch.uint8))
        assert torch.equal(cta.codec.bitmask_decompress(v, bm, w.shape, ro).view(torch.uint8), b)
        if w.numel() <= 1 << 22:
            ov, obm, oro = O.bitmask_compress(w.cpu().view(torch.int8))
            assert torch.equal(v.cpu().view(torch.int8), ov) and torch.equal(bm.cpu(), obm) and torch.equal(ro.cpu(), oro)
    many = cta.codec.bitmask_compress_many(ws)
    for w, (v, bm, ro) in zip(ws, many):
        sv, sbm, sro = cta.codec.bitmask_compress(w)
        assert v.dtype == w.dtype and torch.equal(v.view(torch.uint8), sv.view(torch.uint8)) and torch.equal(bm, sbm) and torch.equal(ro, sro)


def test_bitmask_decompress_many_and_the_batch_entry(cta, dev):
    """Round 6: loading a sparse checkpoint — `codec.bitmask_decompress_many` / `BitmaskCompressor.decompress_state_dict` expand a list of tensors with
    ONE `ct_bitmask_decompress_batch` launch per element size: equal to `bitmask_decompress` tensor by tensor and to the original tensors, for rows of
    one tile, rows of several tiles (> 8192 columns), partial tiles, all-zero / dense tensors, float32, and — taken one by one — 8-bit payloads, a
    tensor without row offsets, a CPU tensor, rows that are not whole 64-byte runs.  The C-ABI entry is also called directly."""
    import ctypes

    from compressed_tensors_amd import _lib
    from compressed_tensors_amd.compressors.sparse.sparse_bitmask import BitmaskCompressor

    g = torch.Generator(device=dev).manual_seed(31)
    specs = [(2048, 2048, BF16, 0.5), (256, 2048, BF16, 0.3), (5632, 2048, F16, 0.5), (64, 8192 + 4096, BF16, 0.6), (3, 32768, BF16, 0.5), (100, 64, BF16, 0.0),
             (33, 4096, BF16, 1.0), (300, 1024, F32, 0.5), (17, 2048 + 32, BF16, 0.5), (128, 256, torch.int8, 0.5), (40, 24, BF16, 0.5),
             (9, 11008, BF16, 0.5), (5, 8192 + 64, F16, 0.3), (3, 28672, BF16, 0.7), (4, 16384, BF16, 0.5),  # long rows: flat tiles unless the row is whole tiles
             (64, 512, BF16, 0.5)]
    ws = []
    for r, c, dt, dens in specs:
        w = torch.randn(r, c, device=dev, generator=g)
        w = (w * 50).to(dt) if dt is torch.int8 else w.to(dt)
        ws.append(w * (torch.rand(r, c, device=dev, generator=g) < dens) if dens < 1.0 else (w.abs() + 1).to(dt))
    comp = [cta.codec.bitmask_compress(w) for w in ws]
    items = [(v, bm, w.shape, ro) for w, (v, bm, ro) in zip(ws, comp)]
    items[-1] = (items[-1][0], items[-1][1], items[-1][2], None)            # no row offsets: the per-tensor path rebuilds them
    items.append((comp[0][0].cpu(), comp[0][1].cpu(), ws[0].shape, comp[0][2].cpu()))  # a CPU entry
    ws = [torch.where(w != 0, w, torch.zeros_like(w)) for w in ws]  # (a dropped -0.0 comes back as +0.0)
    want = ws + [ws[0].cpu()]
    got = cta.codec.bitmask_decompress_many(items)
    for (v, bm, shape, ro), o, w in zip(items, got, want):
        assert o.dtype == w.dtype and o.device == w.device and tuple(o.shape) == tuple(w.shape)
        assert torch.equal(o.view(torch.uint8), w.view(torch.uint8)), (tuple(w.shape), w.dtype)
        assert torch.equal(o.view(torch.uint8), cta.codec.bitmask_decompress(v, bm, shape, ro).view(torch.uint8))
    state = {f"layers.{i}.weight": w for i, w in enumerate(ws)}
    state["norm.bias"] = torch.ones(4, device=dev)
    back = BitmaskCompressor.decompress_state_dict(BitmaskCompressor.compress_state_dict(state))
    assert list(back) == ["norm.bias"] + [f"layers.{i}.weight" for i in range(len(ws))] and all(torch.equal(back[k].view(torch.uint8), state[k].view(torch.uint8)) for k in state)
    # the entry itself, the way a C host calls it
    lib = _lib.load()
    take = [i for i, (r, c, dt, _) in enumerate(specs) if dt is BF16 and (c * 2) % 64 == 0]
    tab = (_lib.BitmaskDItem * len(take))()
    outs = []
    for k, i in enumerate(take):
        v, bm, ro = comp[i]
        o = torch.full_like(ws[i], 7)
        outs.append(o)
        tab[k].values, tab[k].bitmask, tab[k].row_offsets, tab[k].out = v.data_ptr() if v.numel() else None, bm.data_ptr(), ro.data_ptr(), o.data_ptr()
        tab[k].rows, tab[k].cols, tab[k].values_len, tab[k].dt = ws[i].shape[0], ws[i].shape[1], v.numel(), _lib.BF16
    blocks = lib.ct_bitmask_decompress_batch_plan(ctypes.cast(tab, ctypes.c_void_p), len(take))
    assert blocks > 0, _lib.last_error()
    table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
    assert lib.ct_bitmask_decompress_batch(table.data_ptr(), len(take), blocks, 2, _lib.stream_of_device(dev)) == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert all(torch.equal(o.view(torch.int16), ws[i].view(torch.int16)) for o, i in zip(outs, take))
    tab[0].cols = 24  # rows that are not whole 64-byte runs: refused by the plan
    assert lib.ct_bitmask_decompress_batch_plan(ctypes.cast(tab, ctypes.c_void_p), len(take)) == -1 and "not eligible" in _lib.last_error()


def test_c_abi_launches_are_hip_graph_capturable(cta, dev):
    """The compute entries allocate nothing and never synchronise, so a caller can capture a launch-bound sequence of them into a HIP graph and
    replay it: W4 compress + decompress, W3, the sparse decompress, the 2:4 compress and — with its workspace cleared INSIDE the
    graph, because a launch's generation tag is baked in at capture time — the one-pass sparse compress.  Replayed three times on changing
    inputs: every output equal to the eager calls'."""
    from compressed_tensors_amd import _lib

    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(12)
    n, c = 1024, 2048
    w = torch.randn(n, c, device=dev, generator=g).to(BF16)
    sc, zp = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=False)
    sc3, zp3 = cta.codec.minmax_qparams(w, num_bits=3, group_size=128, symmetric=True)
    x = w * (torch.rand(n, c, device=dev, generator=g) < 0.5)
    packed, out = torch.empty(n, c // 8, dtype=torch.int32, device=dev), torch.empty_like(w)
    packed3, out3 = torch.empty(n, c * 3 // 32, dtype=torch.int32, device=dev), torch.empty_like(w)
    vals, bm, ro = torch.empty(n * c, dtype=BF16, device=dev), torch.empty(n, c // 8, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.int64, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    wsb = int(lib.ct_bitmask_compress_workspace_bytes(n, c))
    ws = torch.empty(wsb // 8 + 1, dtype=torch.int64, device=dev)
    dense = torch.empty_like(w)
    v24, bm24 = torch.empty(n, c // 2, dtype=BF16, device=dev), torch.empty(n, c // 8, dtype=torch.uint8, device=dev)
    B, I8 = _lib.BF16, _lib.I8

    def launches(stream):
        rcs = [lib.ct_quant_pack(w.data_ptr(), B, sc.data_ptr(), B, zp.data_ptr(), I8, n, c, 1, 128, c // 128, None, 4, B, packed.data_ptr(), stream),
               lib.ct_unpack_dequant(packed.data_ptr(), n, c // 8, c, 4, sc.data_ptr(), B, zp.data_ptr(), I8, 1, 128, c // 128, None, out.data_ptr(), B, stream),
               lib.ct_quant_pack(w.data_ptr(), B, sc3.data_ptr(), B, None, -1, n, c, 1, 128, c // 128, None, 3, B, packed3.data_ptr(), stream),
               lib.ct_unpack_dequant(packed3.data_ptr(), n, c * 3 // 32, c, 3, sc3.data_ptr(), B, None, -1, 1, 128, c // 128, None, out3.data_ptr(), B, stream)]
        ws.zero_()  # (on the capture stream: a memset node in front of the sparse compress, so that no count word of an earlier replay carries this launch's tag)
        rcs += [lib.ct_bitmask_compress(x.data_ptr(), B, n, c, vals.data_ptr(), n * c, bm.data_ptr(), ro.data_ptr(), total.data_ptr(), ws.data_ptr(), wsb, stream),
                lib.ct_bitmask_decompress(vals.data_ptr(), n * c, bm.data_ptr(), ro.data_ptr(), -1, B, n, c, dense.data_ptr(), stream),
                lib.ct_sparse24_compress(w.data_ptr(), B, n, c, v24.data_ptr(), bm24.data_ptr(), stream)]
        assert not any(rcs), (rcs, _lib.last_error())

    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        launches(_lib.stream_on(dev, side.cuda_stream))  # warm-up outside the capture
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launches(_lib.stream_on(dev, torch.cuda.current_stream(dev).cuda_stream))
    for rep in range(3):
        w.copy_(torch.randn(n, c, device=dev, generator=g).to(BF16))
        x.copy_(w * (torch.rand(n, c, device=dev, generator=g) < 0.2 + 0.3 * rep))
        for t in (packed, out, packed3, out3, vals, bm, ro, dense, v24, bm24):
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        kw = dict(num_bits=4, strategy="group", group_size=128)
        assert torch.equal(packed, cta.codec.quantize_and_pack(w, sc, zp, **kw)) and eq(out.cpu(), cta.codec.unpack_and_dequantize(packed, (n, c), sc, zp, num_bits=4).cpu())
        assert torch.equal(packed3, cta.codec.quantize_and_pack(w, sc3, None, num_bits=3, strategy="group", group_size=128))
        assert eq(out3.cpu(), cta.codec.unpack_and_dequantize(packed3, (n, c), sc3, None, num_bits=3).cpu())
        rv, rbm, rro = cta.codec.bitmask_compress(x)
        nnz = int(total)
        assert nnz == rv.numel() and torch.equal(vals[:nnz], rv) and torch.equal(bm, rbm) and torch.equal(ro, rro), rep
        assert torch.equal(dense.view(torch.int16), torch.where(x != 0, x, torch.zeros_like(x)).view(torch.int16))
        r24 = cta.codec.sparse24_bitmask_compress(w)
        assert torch.equal(v24.view(torch.int16), r24[0].view(torch.int16).reshape(v24.shape)) and torch.equal(bm24, r24[1])


def test_bitmask_compress_batch_through_the_c_abi(cta, dev):
    """`ct_bitmask_batch_plan` + `ct_bitmask_compress_batch` + `ct_copy_batch` called the way a C host would (ctypes structures, no Python
    codec in between): a table of 16-bit tensors of very different sizes — one workgroup, a partial last tile, rows that are not a multiple
    of anything, an all-zero and a dense tensor, 8192 x 8192 (two residency rounds, next to small neighbours) — every output bit-identical
    to `ct_bitmask_compress` item by item and to the CPU oracle; then the float32 table; then a table that mixes element sizes is refused."""
    import ctypes

    from compressed_tensors_amd import _lib

    lib = _lib.load()
    stream = _lib.stream_of_device(dev)
    g = torch.Generator(device=dev).manual_seed(4)
    for dtype in (BF16, F32):
        shapes = [(1, 8), (3, 40), (64, 256), (257, 1000), (2048, 2048), (256, 2048), (5632, 2048), (8192, 8192 if dtype is BF16 else 2048), (100, 64), (33, 4096)]
        xs = []
        for i, (r, c) in enumerate(shapes):
            w = torch.randn(r, c, device=dev, generator=g).to(dtype)
            dens = [0.5, 0.3, 0.0, 0.7, 0.5, 0.9, 0.5, 0.5, 1.0, 0.02][i]
            xs.append(w * (torch.rand(r, c, device=dev, generator=g) < dens) if dens < 1.0 else w.abs() + 1)
        n = len(xs)
        items = (_lib.BitmaskItem * n)()
        totals = torch.full((n,), -7, dtype=torch.int64, device=dev)
        outs = []
        for i, x in enumerate(xs):
            r, c = x.shape
            v, bm, ro = torch.empty(r * c, dtype=dtype, device=dev), torch.empty(r, (c + 7) // 8, dtype=torch.uint8, device=dev), torch.empty(r, dtype=torch.int64, device=dev)
            outs.append((v, bm, ro))
            it = items[i]
            it.x, it.values, it.bitmask, it.row_offsets, it.total = x.data_ptr(), v.data_ptr(), bm.data_ptr(), ro.data_ptr(), totals[i:].data_ptr()
            it.rows, it.cols, it.values_capacity, it.dt = r, c, r * c, _lib.DT[dtype]
        ws_bytes = ctypes.c_int64()
        blocks = lib.ct_bitmask_batch_plan(ctypes.cast(items, ctypes.c_void_p), n, ctypes.byref(ws_bytes))
        assert blocks > 0 and ws_bytes.value > 0, _lib.last_error()
        assert [items[i].first_block for i in range(n)] == [sum(items[j].nwg for j in range(i)) for i in range(n)]
        table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
        ws = torch.empty(ws_bytes.value // 8 + 1, dtype=torch.int64, device=dev)
        for rep in range(3):  # the workspace is reused as it is: every launch has its own generation tags
            if rep:
                blocks = lib.ct_bitmask_batch_plan(ctypes.cast(items, ctypes.c_void_p), n, ctypes.byref(ws_bytes))
                table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)
            totals.fill_(-7)
            rc = lib.ct_bitmask_compress_batch(table.data_ptr(), n, blocks, xs[0].element_size(), ws.data_ptr(), ws_bytes.value, stream)
            assert rc == 0, _lib.last_error()
            torch.cuda.synchronize()
            for i, (x, (v, bm, ro)) in enumerate(zip(xs, outs)):
                rv, rbm, rro = cta.codec.bitmask_compress(x)
                nnz = int(totals[i])
                assert nnz == rv.numel(), (i, nnz, rv.numel())
                assert torch.equal(v[:nnz], rv) and torch.equal(bm, rbm) and torch.equal(ro, rro), (dtype, i)
        # the oracle on the items it finishes quickly
        for i in (1, 3, 5, 9):
            ov, obm, oro = O.bitmask_compress(xs[i].cpu())
            assert torch.equal(outs[i][0][: ov.numel()].cpu().view(torch.uint8), ov.view(torch.uint8)) and torch.equal(outs[i][1].cpu(), obm) and torch.equal(outs[i][2].cpu(), oro)
        # the batched copy: every kept prefix into its own exact-size buffer
        cp = (_lib.CopyItem * n)()
        exact = [torch.empty(int(totals[i]), dtype=dtype, device=dev) for i in range(n)]
        for i in range(n):
            cp[i].src, cp[i].dst, cp[i].bytes = outs[i][0].data_ptr(), exact[i].data_ptr(), exact[i].numel() * exact[i].element_size()
        cblocks = lib.ct_copy_batch_plan(ctypes.cast(cp, ctypes.c_void_p), n)
        assert cblocks >= 0
        ctab = torch.frombuffer(bytearray(bytes(cp)), dtype=torch.uint8).to(dev)
        assert lib.ct_copy_batch(ctab.data_ptr(), n, cblocks, stream) == 0
        torch.cuda.synchronize()
        assert all(torch.equal(e, o[0][: e.numel()]) for e, o in zip(exact, outs))
    # a misaligned source / destination / odd byte count takes the byte path of the copy kernel
    src = torch.arange(100000, dtype=torch.int32, device=dev).view(torch.uint8)
    dst = torch.zeros(100000 * 4 + 64, dtype=torch.uint8, device=dev)
    one = (_lib.CopyItem * 2)()
    one[0].src, one[0].dst, one[0].bytes = src.data_ptr() + 3, dst.data_ptr() + 5, 70001
    one[1].src, one[1].dst, one[1].bytes = src.data_ptr() + 160000, dst.data_ptr() + 160000, 33333
    cb = lib.ct_copy_batch_plan(ctypes.cast(one, ctypes.c_void_p), 2)
    tab = torch.frombuffer(bytearray(bytes(one)), dtype=torch.uint8).to(dev)
    assert lib.ct_copy_batch(tab.data_ptr(), 2, cb, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst[5:70006], src[3:70004]) and torch.equal(dst[160000:193333], src[160000:193333]) and int(dst[70006:160000].sum()) == 0
    # one element size per table
    mixed = (_lib.BitmaskItem * 2)()
    for i, x in enumerate((xs[2], xs[2].to(BF16))):
        v, bm, ro = torch.empty(x.numel(), dtype=x.dtype, device=dev), torch.empty(x.shape[0], x.shape[1] // 8, dtype=torch.uint8, device=dev), torch.empty(x.shape[0], dtype=torch.int64, device=dev)
        mixed[i].x, mixed[i].values, mixed[i].bitmask, mixed[i].row_offsets, mixed[i].total = x.data_ptr(), v.data_ptr(), bm.data_ptr(), ro.data_ptr(), totals.data_ptr()
        mixed[i].rows, mixed[i].cols, mixed[i].values_capacity, mixed[i].dt = x.shape[0], x.shape[1], x.numel(), _lib.DT[x.dtype]
    assert lib.ct_bitmask_batch_plan(ctypes.cast(mixed, ctypes.c_void_p), 2, ctypes.byref(ws_bytes)) == -1 and "ONE element size" in _lib.last_error()


@pytest.mark.parametrize("exact", [True, False], ids=["exact", "view"])
def test_from_dense_many_equals_from_dense(cta, dev, exact):
    """Round 6: a list of tensors through `BitmaskTensor.from_dense_many` — windows of compress launches, one mailbox
    word each, ONE host wait per window — gives exactly what `from_dense` gives tensor by tensor: a TinyLlama layer's shapes at several
    densities, a float32 and an int8-viewed fp8 tensor, a CPU tensor and a transposed view (both taken one by one), an empty list; a small
    arena budget (every window one tensor) and more tensors than mailbox words (several windows)"""
    from compressed_tensors_amd.compressors.sparse.sparse_bitmask import BitmaskCompressor, BitmaskTensor

    g = torch.Generator(device=dev).manual_seed(21)
    shapes = [(2048, 2048), (256, 2048), (256, 2048), (2048, 2048), (5632, 2048), (5632, 2048), (2048, 5632)]
    ws = []
    for rep in range(10):  # 70 tensors: more than the 56 words of a window
        for i, (r, c) in enumerate(shapes):
            if rep > 1 and r * c > 1 << 22:
                r //= 8
            w = torch.randn(r, c, device=dev, generator=g).to(BF16)
            ws.append(w * (torch.rand(r, c, device=dev, generator=g) < (0.1 + 0.1 * ((i + rep) % 9))))
    ws.append(torch.randn(300, 1000, device=dev, generator=g) * (torch.rand(300, 1000, device=dev, generator=g) < 0.5))  # float32
    ws.append((torch.randn(128, 256, device=dev, generator=g) * (torch.rand(128, 256, device=dev, generator=g) < 0.5)).to(torch.float8_e4m3fn))
    ws.append(ws[0].cpu())
    ws.append(ws[1].t())
    assert BitmaskTensor.from_dense_many([]) == []
    for budget in (None, 1):
        if budget is None:
            many = BitmaskTensor.from_dense_many(ws, exact=exact)
        else:
            parts = cta.codec.bitmask_compress_many(ws[:9], exact=exact, arena_bytes=budget)
            many = [BitmaskTensor(shape=t.shape, compressed=v, bitmask=b, row_offsets=r) for t, (v, b, r) in zip(ws[:9], parts)]
        for w, bt in zip(ws, many):
            one = BitmaskTensor.from_dense(w, exact=exact)
            assert bt.shape == one.shape and bt.compressed.dtype == w.dtype and bt.compressed.device == w.device
            assert torch.equal(bt.compressed.view(torch.uint8), one.compressed.view(torch.uint8)) and torch.equal(bt.bitmask, one.bitmask) and torch.equal(bt.row_offsets, one.row_offsets)
            if w.is_cuda and w.is_contiguous():
                nnz_bytes = int((w.float() != 0).sum()) * w.element_size()
                assert bt.compressed.untyped_storage().nbytes() <= (nnz_bytes + (2 << 20) if exact else w.numel() * w.element_size())
            assert torch.equal(bt.decompress().float(), w.float())
    # and the state-dict interface on top of it
    state = {f"model.layers.{i}.w.weight": w for i, w in enumerate(ws[:12])}
    state["model.norm.bias"] = torch.ones(8, device=dev)
    comp = BitmaskCompressor.compress_state_dict(state)
    assert "model.norm.bias" in comp and "model.layers.3.w.compressed" in comp and "model.layers.3.w.weight" not in comp
    back = BitmaskCompressor.decompress_state_dict(comp)
    assert sorted(back) == sorted(state) and all(torch.equal(back[k].float(), state[k].float()) for k in state)


def test_waiting_calls_native_host_path_equals_the_python_host_path(cta, dev):
    """the two plug-in calls that wait for the device — sparse-bitmask compress (nnz) and the default mode of marlin-24 compress (the
    2:4 verdict) — run their host side in csrc/host/ct_hostpath.cpp when the extension is built; the Python host side stays for what
    the extension declines (and for a build without it).  Same kernels either way: every output equal, the same ValueError, and the
    native path really is the one the default call takes (counted)."""
    from compressed_tensors_amd import _lib as ctlib

    hp = ctlib.hostpath()
    assert hp is not None, "compressed_tensors_amd/_hostpath.so is missing"
    taken = {"bitmask_compress": 0, "marlin24_compress_default": 0}

    class Counting:
        def __getattr__(self, name):
            fn = getattr(hp, name)
            if name in taken:
                def counted(*a, _fn=fn, _n=name):
                    r = _fn(*a)
                    taken[_n] += r is not None
                    return r
                return counted
            return fn

    g = torch.Generator().manual_seed(5)
    dense = torch.randn(512, 1024, generator=g).to(BF16)
    cases = [dense * (torch.rand(512, 1024, generator=g) < d) for d in (0.5, 0.1, 0.0, 1.0)]
    cases += [cases[0].to(F16), cases[0].float(), cases[0][:, :1000].contiguous(), cases[0].t()]  # .t(): declined by the native path
    w24 = torch.randn(256, 1024, generator=g).to(BF16)
    w24 = w24 * O.sparse24_mask(w24).to(BF16)
    scale, zp = O.calculate_qparams_minmax(w24.to(F16), num_bits=4, group_size=128, symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, strategy="group", group_size=128, symmetric=True))
    sd = {"weight": w24.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}
    bad = dict(sd, weight=torch.randn(256, 1024, generator=g).to(BF16).to(dev))

    def run():
        out = [cta.codec.bitmask_compress(c.to(dev)) for c in cases]
        got = cta.Marlin24Compressor.compress(sd, scheme)
        out.append(tuple(got[k] for k in ("weight_packed", "scale_packed", "meta")))
        with pytest.raises(ValueError, match="2:4 sparsity structure"):
            cta.Marlin24Compressor.compress(bad, scheme)
        return out

    ctlib._HOSTPATH[0] = Counting()
    try:
        native = run()
        ctlib._HOSTPATH[0] = None
        python = run()
    finally:
        ctlib._HOSTPATH[0] = hp
    assert taken == {"bitmask_compress": len(cases) - 1, "marlin24_compress_default": 2}, taken
    for a, b in zip(native, python):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and x.shape == y.shape and x.device == y.device and torch.equal(x.view(torch.uint8) if x.is_floating_point() else x,
                                                                                                       y.view(torch.uint8) if y.is_floating_point() else y)
    # and against the checker
    v, bm, ro = native[0]
    rv, rbm, rro = O.bitmask_compress(cases[0])
    assert torch.equal(v.cpu().view(torch.int16), rv.view(torch.int16)) and torch.equal(bm.cpu(), rbm) and torch.equal(ro.cpu(), rro)


def test_waiting_calls_from_several_threads(cta, dev):
    """the calls that wait for the device drop the GIL while they spin, so Python threads really are inside them at the same time: every
    thread has its own mailbox word (`_lib.mailbox`: per thread and device), every call its own workspace, the launches share the default
    stream — four threads x (sparse-bitmask compress of its own tensors, marlin-24 compress in default mode with and without a 2:4
    violation), every result equal to what one thread gets"""
    import threading

    g = torch.Generator().manual_seed(9)
    sets = []
    for t in range(4):
        ws = [(torch.randn(256 + 64 * t, 512, generator=g) * (torch.rand(256 + 64 * t, 512, generator=g) < 0.3 + 0.15 * t)).to(BF16).to(dev) for _ in range(3)]
        w24 = torch.randn(128, 512, generator=g).to(BF16)
        w24 = w24 * O.sparse24_mask(w24).to(BF16)
        scale, zp = O.calculate_qparams_minmax(w24.to(F16), num_bits=4, group_size=128, symmetric=True)
        sets.append((ws, {"weight": w24.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}))
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, strategy="group", group_size=128, symmetric=True))
    dense = {"weight": torch.randn(128, 512, generator=g).to(BF16).to(dev), "weight_scale": sets[0][1]["weight_scale"], "weight_zero_point": sets[0][1]["weight_zero_point"]}

    def work(t, rounds):
        ws, sd = sets[t]
        out = []
        for r in range(rounds):
            out.append([tuple(x.clone() for x in cta.codec.bitmask_compress(w)) for w in ws])
            got = cta.Marlin24Compressor.compress(sd, scheme)
            out.append([(got["weight_packed"].clone(), got["scale_packed"].clone(), got["meta"].clone())])
            if (r + t) % 2:
                with pytest.raises(ValueError, match="2:4 sparsity structure"):
                    cta.Marlin24Compressor.compress(dense, scheme)
        return out

    want = [work(t, 1) for t in range(4)]
    results, errors = [None] * 4, []

    def run(t):
        try:
            torch.cuda.set_device(dev)
            results[t] = work(t, 12)
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        for k, group in enumerate(results[t]):
            ref = want[t][k % 2]
            for a, b in zip(group, ref):
                for x, y in zip(a, b):
                    assert x.shape == y.shape and torch.equal(x.view(torch.uint8) if x.is_floating_point() else x, y.view(torch.uint8) if y.is_floating_point() else y), (t, k)


@pytest.mark.parametrize("shape", [(192, 1024), (64, 288), (128, 256)])
@pytest.mark.parametrize("wdt", [BF16, F16])
def test_marlin24_fused_front_end_vs_unfused(cta, dev, wdt, shape):
    """ct_marlin24_quant_compress == weight.to(fp16) -> quantize (fp16) -> cutlass 2:4 compress
    (k % 256 == 0: tiled kernel with LDS-ordered metadata; otherwise the lane-per-word kernel)"""
    g = torch.Generator().manual_seed(3)
    w = torch.randn(shape, generator=g).to(wdt)
    w = (w * O.sparse24_mask(w).to(w.dtype)).to(dev)
    w[0, :4] = 0
    gs = 128 if shape[1] % 128 == 0 else 32
    scale, zp = cta.codec.minmax_qparams(w.to(F16), num_bits=4, group_size=gs, symmetric=True)
    comp, meta, bad = cta.codec.marlin24_quant_compress(w, scale, zp, num_bits=4, group_size=gs)
    q = cta.codec.quantize_tensor(w.to(F16), scale, zp, num_bits=4, strategy="group", group_size=gs)
    comp_ref, meta_ref = cta.codec.cutlass24_from_dense(q)
    assert int(bad.item()) == 0
    assert torch.equal(comp.cpu().float(), comp_ref.cpu().float()) and torch.equal(meta.cpu(), meta_ref.cpu())


_BITMASK_FORMS_CODE = r"""
import sys, torch
sys.path.insert(0, %r)
from compressed_tensors_amd import _lib
_lib.LIB_PATH = _lib.DIAG_LIB_PATH  # the CT_BITMASK_RESIDENT* knobs exist in the diagnostics build only (-DCT_DIAG)
from compressed_tensors_amd import codec
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
for (r, c, dens, dt) in ((4096, 4096, 0.5, torch.bfloat16), (1024, 2048, 0.1, torch.float16), (3000, 1000, 0.9, torch.bfloat16), (513, 8200, 0.5, torch.bfloat16),
                         (2048, 4096, 0.0, torch.bfloat16), (1024, 4096, 1.0, torch.float16), (8192, 8192, 0.5, torch.bfloat16), (7, 8, 0.5, torch.int16),
                         (1, 8, 1.0, torch.bfloat16), (4099, 24, 0.5, torch.float16),
                         # 32-bit payloads ride the resident kernel as pairs of halves (one mask bit per element, offsets and totals halved)
                         (4096, 4096, 0.5, torch.float32), (513, 8200, 0.5, torch.float32), (3000, 1000, 0.9, torch.float32), (2048, 4104, 0.3, torch.float32),
                         (1024, 4096, 0.0, torch.float32), (512, 4096, 1.0, torch.float32), (7, 8, 0.5, torch.int32), (1, 8, 1.0, torch.float32),
                         (4099, 24, 0.5, torch.float32), (4096, 8192, 0.5, torch.float32), (1024, 2048, 0.5, torch.int32),
                         (37, 8208, 0.5, torch.float32), (5, 4112, 0.7, torch.float32), (64, 48, 0.5, torch.float32)):  # mask rows of 2 (mod 4) bytes
    w = torch.randn(r, c, device=dev, generator=g)
    w = w.masked_fill(torch.rand(r, c, device=dev, generator=g) >= dens, 0)
    w = (w * 100).to(dt) if dt in (torch.int16, torch.int32) else w.to(dt)
    if dt not in (torch.int16, torch.int32) and dens > 0:
        w.view(-1)[::7] = -0.0  # a negative zero is a zero
    if dt == torch.float32 and dens > 0:
        w.view(torch.int32).view(-1)[3::11] = 0x00010000  # a denormal whose low half is zero: non-zero as an element, zero as a half
        w.view(torch.int32).view(-1)[5::13] = 0x00000001  # ... and one whose high half is zero
    for rep in range(3):  # a fresh generation tag per call over recycled workspace memory
        v, bm, ro = codec.bitmask_compress(w)
        v2, bm2, ro2 = codec.bitmask_compress(w, two_pass=True)
        assert v.numel() == v2.numel() and torch.equal(v.view(torch.uint8), v2.view(torch.uint8)) and torch.equal(bm, bm2) and torch.equal(ro, ro2), (r, c, dens, dt, rep)
    back = codec.bitmask_decompress(v, bm, (r, c), ro)
    expect = torch.where(w != 0, w, torch.zeros_like(w))  # -0.0 comes back as +0.0
    assert torch.equal(back.view(torch.uint8), expect.view(torch.uint8)) and torch.equal(codec.bitmask_decompress(v, bm, (r, c)).view(torch.uint8), expect.view(torch.uint8)), (r, c, dens, dt)
print("FORMS_OK")
"""


@pytest.mark.parametrize("env", [
    {},                                          # the resident form (default)
    {"CT_BITMASK_RESIDENT": "0"},                # the two kernels (count + scatter)
    {"CT_BITMASK_RESIDENT_MAX_WGS": "3"},        # resident, the tensor in chunks of three workgroups with a running total between them
    {"CT_BITMASK_RESIDENT_WAIT_US": "0"},        # no waiting at all: every workgroup recounts the shares whose counts have not arrived (self-help)
    {"CT_BITMASK_RESIDENT_WAIT_US": "0", "CT_BITMASK_RESIDENT_MAX_WGS": "5"},
], ids=["resident", "two_kernels", "resident_chunked", "resident_self_help", "resident_self_help_chunked"])
def test_bitmask_compress_forms_agree(env):
    """every form of the 16- / 32-bit sparse compress (the knobs are read once per process, hence the subprocess): values, bitmask, row
    offsets and the total bit-identical to count / scan / scatter for ragged, empty, dense, tiny shapes"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _BITMASK_FORMS_CODE % root], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "FORMS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype,shapes", [
    (BF16, [(2048, 4096), (2048, 4096 + 8), (4096, 4096), (4100, 4096), (4096, 6144), (6000, 4096), (8192, 5120), (8192, 8192)]),
    (F32, [(2048, 2048), (2049, 2048), (4096, 2048), (4096, 3072), (3000, 4096), (4096, 4096)]),
    (torch.int8, [(4096, 4096), (8192, 8192), (8192, 8192 + 16), (3, 16)]),
], ids=["bf16", "fp32", "int8"])
def test_bitmask_compress_every_tiles_per_wave_instantiation(cta, dev, dtype, shapes):
    """Round 6: the resident compress is instantiated per tiles-per-wave (1, 2, 3, 4 — the launch picks the one that matches the tensor's size, so that a
    wave loads no tile it does not use) and 8-bit payloads take one tile per wave at four workgroups per CU.  Sizes on both sides of every boundary,
    against eager torch on the device: values, bitmask, row offsets, total."""
    g = torch.Generator(device=dev).manual_seed(61)
    weights = (1 << torch.arange(8, device=dev, dtype=torch.int32))
    for (r, c) in shapes:
        if dtype is torch.int8:
            w = torch.randint(-127, 128, (r, c), device=dev, generator=g, dtype=torch.int16).to(torch.int8)
        else:
            w = torch.randn(r, c, device=dev, generator=g, dtype=dtype)
        w = w.masked_fill_(torch.rand(r, c, device=dev, generator=g) < 0.5, 0)
        v, bm, ro = cta.codec.bitmask_compress(w)
        m = w != 0
        cnt = m.sum(-1)
        assert v.numel() == int(cnt.sum()) and torch.equal(v, w[m]), (r, c)
        assert torch.equal(ro, torch.cumsum(cnt, 0) - cnt), (r, c)
        assert torch.equal(bm, (m.view(r, c // 8, 8).to(torch.int32) * weights).sum(-1).to(torch.uint8)), (r, c)
        assert torch.equal(cta.codec.bitmask_decompress(v, bm, w.shape, ro), w), (r, c)
        del w, v, bm, ro, m


@pytest.mark.parametrize("dtype", [BF16, F16, F32, torch.int16, torch.int32], ids=["bf16", "fp16", "fp32", "int16", "int32"])
def test_bitmask_compress_every_bit_pattern(cta, dev, dtype):
    """the non-zero test of the sparse compress on every 16-bit pattern (and, for 32-bit payloads, every exponent x sign x {zero, lowest bit,
    highest bit, all ones} significand): +-0 are zeros, denormals, NaNs and infinities are kept, integers by value — as `x != 0` in torch;
    round 5's row form (32-bit payloads) and the unit form (16-bit) against the integer rule, values / bitmask / row offsets bit for bit"""
    import numpy as np

    if dtype.itemsize == 2:
        bits = torch.arange(65536, dtype=torch.int32).to(torch.int16).reshape(128, 512)
        x = bits.view(dtype)
    else:
        sign, expo = torch.arange(2, dtype=torch.int64), torch.arange(256, dtype=torch.int64)
        frac = torch.tensor([0, 1, 1 << 22, (1 << 23) - 1, 0x2aaaaa, 12345, 1 << 12, 0x7ffffe], dtype=torch.int64)
        bits = (sign[:, None, None] << 31) | (expo[None, :, None] << 23) | frac[None, None, :]
        bits = bits.reshape(-1).repeat(4)  # 16384 elements
        bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32).reshape(64, 256)
        x = bits.view(dtype)
    is_float = dtype in (BF16, F16, F32)
    mask_sign = (1 << (8 * dtype.itemsize - 1)) - 1
    keep = ((bits.to(torch.int64) & mask_sign) != 0) if is_float else (bits != 0)
    assert torch.equal(keep, x != 0) or not is_float  # torch agrees with the integer rule (float `!=`: NaN != 0 is True, -0.0 != 0 is False)
    iv = torch.int16 if dtype.itemsize == 2 else torch.int32
    want_v = bits.to(iv)[keep]
    want_bm = torch.from_numpy(np.packbits(keep.numpy(), axis=1, bitorder="little"))
    want_ro = torch.cumsum(keep.sum(1), 0) - keep.sum(1)
    for two_pass in (False, True):
        v, bm, ro = cta.codec.bitmask_compress(x.to(dev), two_pass=two_pass)
        assert v.dtype == dtype and torch.equal(v.cpu().view(iv), want_v) and torch.equal(bm.cpu(), want_bm) and torch.equal(ro.cpu(), want_ro), (dtype, two_pass)
    back = cta.codec.bitmask_decompress(v, bm, x.shape, ro)
    want_back = torch.where(keep, bits.to(iv), torch.zeros_like(bits.to(iv)))  # a dropped -0.0 comes back as +0.0
    assert torch.equal(back.cpu().view(iv), want_back)


def test_bitmask_compress_concurrent_streams(cta, dev):
    """resident compress kernels in flight at once on two streams (raw ABI, no host synchronisation between launches) next to a GEMM that
    occupies CUs: workgroups of one kernel wait for workgroups that are not resident yet.  Dependencies only point to earlier workgroups,
    so every launch must finish with the right total — results bit-identical to count / scan / scatter"""
    from compressed_tensors_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(23)
    ws, ref, bufs = [], [], []
    for n in (4096, 3072):
        w = torch.randn(n, n, dtype=BF16, device=dev, generator=g)
        w = w.masked_fill(torch.rand(n, n, device=dev, generator=g) < 0.5, 0)
        ws.append(w)
        ref.append(cta.codec.bitmask_compress(w, two_pass=True))
        nb = int(lib.ct_bitmask_compress_workspace_bytes(n, n))
        reps = []
        for rep in range(5):
            reps.append(dict(wk=torch.empty(nb // 8 + 1, dtype=torch.int64, device=dev), vals=torch.empty(n * n, dtype=BF16, device=dev),
                             bm=torch.empty(n, n // 8, dtype=torch.uint8, device=dev), ro=torch.empty(n, dtype=torch.int64, device=dev), nb=nb))
        bufs.append(reps)
    a = torch.randn(4096, 4096, dtype=BF16, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    for rep in range(5):
        with torch.cuda.stream(streams[2]):
            a @ a
        for k in (0, 1):
            b, w, n = bufs[k][rep], ws[k], ws[k].shape[0]
            rc = lib.ct_bitmask_compress(w.data_ptr(), _lib.BF16, n, n, b["vals"].data_ptr(), b["vals"].numel(), b["bm"].data_ptr(), b["ro"].data_ptr(),
                                         b["wk"][-1:].data_ptr(), b["wk"].data_ptr(), b["nb"], streams[k].cuda_stream)
            assert rc == 0
    torch.cuda.synchronize()
    for k in (0, 1):
        v_ref, bm_ref, ro_ref = ref[k]
        for b in bufs[k]:
            nnz = int(b["wk"][-1].item())
            assert nnz == v_ref.numel()
            assert torch.equal(b["vals"][:nnz].view(torch.int16), v_ref.view(torch.int16)) and torch.equal(b["bm"], bm_ref) and torch.equal(b["ro"], ro_ref)


def test_bitmask_resident_self_help_total(dev):
    """the raw ABI with a zero wait budget: workgroups recount what they would have waited for; *total is the right count"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from compressed_tensors_amd import _lib
_lib.LIB_PATH = _lib.DIAG_LIB_PATH  # CT_BITMASK_RESIDENT_WAIT_US exists in the diagnostics build only (-DCT_DIAG)
lib = _lib.load(); dev = torch.device("cuda:0")
N = 4096
w = torch.randn(N, N, dtype=torch.bfloat16, device=dev)
w = w.masked_fill(torch.rand(N, N, device=dev) < 0.37, 0)
ws_bytes = int(lib.ct_bitmask_compress_workspace_bytes(N, N))
wk = torch.zeros(ws_bytes // 8 + 1, dtype=torch.int64, device=dev)
vals = torch.empty(N * N, dtype=torch.bfloat16, device=dev); bm = torch.empty(N, N // 8, dtype=torch.uint8, device=dev); ro = torch.empty(N, dtype=torch.int64, device=dev)
rc = lib.ct_bitmask_compress(w.data_ptr(), _lib.BF16, N, N, vals.data_ptr(), vals.numel(), bm.data_ptr(), ro.data_ptr(), wk[-1:].data_ptr(), wk.data_ptr(), ws_bytes,
                             torch.cuda.current_stream(dev).cuda_stream)
torch.cuda.synchronize()
nnz = int((w != 0).sum().item())
print("TOTAL", rc, int(wk[-1].item()) == nnz, bool(torch.equal(vals[:nnz], w[w != 0])))
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, CT_BITMASK_RESIDENT_WAIT_US="0"))
    assert r.returncode == 0 and "TOTAL 0 True True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


# ----------------------------------------------------------------------------- modules / staging
def test_compress_decompress_module(cta, dev):
    """reference tests/test_compressors/test_compress_decompress_module.py:24-127"""
    torch.manual_seed(0)
    lin = torch.nn.Linear(256, 256, bias=True).to(dev).to(BF16)
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=False)
    lin.quantization_scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    scale, zp = cta.quantization.calculate_qparams_from_weight(lin.weight.data, args)
    lin.register_parameter("weight_scale", torch.nn.Parameter(scale, requires_grad=False))
    lin.register_parameter("weight_zero_point", torch.nn.Parameter(zp, requires_grad=False))
    w0 = lin.weight.data.clone()
    cta.compress_module(lin)
    assert lin.quantization_status == cta.QuantizationStatus.COMPRESSED
    names = dict(lin.named_parameters())
    assert "weight" not in names and names["weight_packed"].dtype == torch.int32 and not names["weight_packed"].requires_grad
    assert names["weight_zero_point"].dtype == torch.int32 and names["weight_shape"].tolist() == [256, 256]
    assert lin.quantization_scheme.format == cta.CompressionFormat.pack_quantized
    cta.decompress_module(lin)
    assert lin.quantization_status == cta.QuantizationStatus.DECOMPRESSED
    assert lin.weight.dtype == BF16 and lin.weight.shape == (256, 256) and lin.weight_zero_point.dtype == torch.int8
    fq = O.fake_quantize(w0.cpu(), scale.cpu(), zp.cpu(), num_bits=4, strategy="group", group_size=128)
    assert torch.equal(lin.weight.data.cpu(), fq)


def test_model_compressor_roundtrip(cta, dev):
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 512, bias=False), torch.nn.ReLU(), torch.nn.Linear(512, 128, bias=False)).to(dev).to(BF16)
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True)
    expect = {}
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.Linear):
            m.quantization_scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
            s, z = cta.quantization.calculate_qparams_from_weight(m.weight.data, args)
            m.register_parameter("weight_scale", torch.nn.Parameter(s, requires_grad=False))
            m.register_parameter("weight_zero_point", torch.nn.Parameter(z, requires_grad=False))
            expect[name] = O.fake_quantize(m.weight.data.cpu(), s.cpu(), z.cpu(), num_bits=4, strategy="group", group_size=128)
    mc = cta.ModelCompressor()
    mc.compress_model(model)
    assert all(hasattr(m, "weight_packed") for m in model if isinstance(m, torch.nn.Linear))
    y = model(torch.randn(4, 256, device=dev, dtype=BF16))  # first forward decompresses (hook)
    assert y.shape == (4, 128) and not hasattr(model, "ct_decompress_hook")
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.Linear):
            assert torch.equal(m.weight.data.cpu(), expect[name])


def test_host_tensors_are_staged_through_the_gpu(cta, dev):
    g = torch.Generator().manual_seed(2)
    v = torch.randint(-8, 8, (32, 200), dtype=torch.int8, generator=g)
    p = cta.codec.pack_to_int32(v, 4)
    assert p.device.type == "cpu" and torch.equal(p, O.pack_to_int32(v, 4).contiguous())


@pytest.mark.parametrize("symmetric", [True, False])
def test_batched_model_compress_matches_per_module(cta, dev, symmetric):
    """ModelCompressor turns the pack-quantized modules into ONE launch per direction
    (ct_quant_pack_batch / ct_unpack_dequant_batch); every state dict must be identical to the
    per-module path, including modules the batch cannot take (ragged columns, fp32)."""
    import copy

    torch.manual_seed(5)
    shapes = [(256, 2048), (64, 256), (2048, 256), (96, 128), (32, 40 * 4), (128, 384)]
    model = torch.nn.Sequential(*[torch.nn.Linear(c, r, bias=False) for r, c in shapes]).to(dev).to(BF16)
    model[5] = model[5].float()  # fp32 weights: not eligible for the batch
    for i, m in enumerate(model):
        gs = 32 if shapes[i][1] % 128 else 128
        if i == 4:
            gs = 40  # group size not a multiple of 32: per-module path
        args = cta.QuantizationArgs(num_bits=4, group_size=gs, symmetric=symmetric, strategy="group")
        if i == 3:
            args = cta.QuantizationArgs(num_bits=4, symmetric=symmetric, strategy="channel")
        m.quantization_scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
        s, z = cta.quantization.calculate_qparams_from_weight(m.weight.data, args)
        m.register_parameter("weight_scale", torch.nn.Parameter(s, requires_grad=False))
        m.register_parameter("weight_zero_point", torch.nn.Parameter(z, requires_grad=False))
    ref = copy.deepcopy(model)
    for m in ref:
        cta.compress_module(m)
    # the C++ host loop (csrc/host/ct_hostpath.cpp) must be built on the GPU box and must be what takes the plain symmetric modules
    from compressed_tensors_amd.compressors.pack_quantized import base as pq

    hp = pq._hostpath()
    assert hp is not None, "compressed_tensors_amd/_hostpath.so is missing: the batched module paths fell back to the Python loop"
    taken, zp_tables = {"compress": 0, "decompress": 0}, {"compress": 0, "decompress": 0}

    class Counting:
        def __getattr__(self, name):
            fn = getattr(hp, name)
            if name in ("w4_plan_compress", "w4_plan_decompress"):
                def counted(modules, infos, _fn=fn, _k=name.rsplit("_", 1)[1]):
                    planned, rest = _fn(modules, infos)
                    taken[_k] += sum(t[1] for t in planned.values())
                    zp_tables[_k] += sum(t[4] for t in planned.values())
                    return planned, rest
                return counted
            return fn

    from compressed_tensors_amd import _lib as ctlib

    ctlib._HOSTPATH[0] = Counting()
    try:
        cta.ModelCompressor().compress_model(model)
        for a, b in zip(model, ref):
            sa, sb = dict(a.named_parameters()), dict(b.named_parameters())
            assert sa.keys() == sb.keys() and "weight_packed" in sa
            assert list(a._parameters) == list(b._parameters)  # the same order too
            for k in sa:
                assert sa[k].dtype == sb[k].dtype and sa[k].device == sb[k].device and torch.equal(sa[k], sb[k]), k
        for m in ref:
            cta.decompress_module(m)
        cta.ModelCompressor().decompress_model(model)
    finally:
        ctlib._HOSTPATH[0] = hp
    # modules 0-3 are plain int4 group / channel modules -> the C++ loop, symmetric or not.  Round 6: an asymmetric scheme's zero points are packed by
    # the weights' own launch (no second table on the compress side); on the decompress side they ride in the weights' launch when the kernel can read
    # the stored form (groups of 128, cols % 512 == 0: module 0) and through the zero-point table otherwise (modules 1-3)
    assert taken == {"compress": 4, "decompress": 4}, taken
    assert zp_tables == ({"compress": 0, "decompress": 0} if symmetric else {"compress": 0, "decompress": 3}), zp_tables
    for a, b in zip(model, ref):
        assert eq(a.weight.data.cpu(), b.weight.data.cpu()) and a.weight.dtype == b.weight.dtype


@pytest.mark.parametrize("kind", ["int8", "fp8", "fp8-zp", "int8-asym", "int6"])
def test_batched_8bit_model_compress_matches_per_module(cta, dev, kind):
    """ModelCompressor with int-quantized / float-quantized / naive-quantized modules: ONE ct_q8_quant_batch / ct_q8_dequant_batch
    launch per direction (model_compressor.py:167-169,196-198 loops); every state dict identical to the per-module path,
    including modules the batch cannot take (ragged columns, fp32 weights, activation ordering is covered elsewhere)"""
    import copy

    torch.manual_seed(7)
    shapes = [(256, 2048), (64, 256), (2048, 256), (96, 128), (32, 40), (128, 384), (48, 512)]
    model = torch.nn.Sequential(*[torch.nn.Linear(c, r, bias=(i == 1)) for i, (r, c) in enumerate(shapes)]).to(dev).to(F16 if kind == "int6" else BF16)
    model[5] = model[5].float()  # fp32 weights: per module
    fp8 = kind in ("fp8", "fp8-zp")  # "fp8-zp": the calibrated flow's float8 zero points are on the modules (batch kind "fp8z", round 6)
    sym = kind != "int8-asym"
    bits = 6 if kind == "int6" else 8
    for i, m in enumerate(model):
        strategy, gs = ("tensor", None) if i == 0 else ("channel", None) if i in (1, 3, 4) else ("group", 128 if i != 6 else 32)
        args = cta.QuantizationArgs(num_bits=bits, type="float" if fp8 else "int", strategy=strategy, group_size=gs, symmetric=sym)
        act = cta.QuantizationArgs(num_bits=8, type="float" if fp8 else "int", strategy="tensor")
        # weight-only INT would infer pack-quantized: the 6-bit case names naive-quantized explicitly (codes in [-32, 31] stored as int8)
        m.quantization_scheme = cta.QuantizationScheme(targets=["Linear"], weights=args, input_activations=None if kind == "int6" else act,
                                                       format="naive-quantized" if kind == "int6" else None)
        s, z = cta.quantization.calculate_qparams_from_weight(m.weight.data, args)
        m.register_parameter("weight_scale", torch.nn.Parameter(s, requires_grad=False))
        if not fp8:
            m.register_parameter("weight_zero_point", torch.nn.Parameter(z, requires_grad=False))
        elif kind == "fp8-zp":
            assert z.dtype == torch.float8_e4m3fn
            if i == 2:  # not all zeros: the kernels must read the zero points as float8 values, not as bytes
                z = torch.full(z.shape, 0.5, device=z.device).to(torch.float8_e4m3fn)
            m.register_parameter("weight_zero_point", torch.nn.Parameter(z, requires_grad=False))
            m.weight.data.view(-1)[:7] = torch.tensor([-0.0, 0.0, -1e-30, 1e-30, -0.0, 3.0, -3.0], dtype=m.weight.dtype, device=dev)  # -0 + zero point = +0
    ref = copy.deepcopy(model)
    for m in ref:
        cta.compress_module(m)
    comp = cta.ModelCompressor()
    comp.compress_model(model)
    fmt = {"int8": "int-quantized", "int8-asym": "int-quantized", "fp8": "float-quantized", "fp8-zp": "float-quantized", "int6": "naive-quantized"}[kind]
    for a, b in zip(model, ref):
        assert enum_name(a.quantization_scheme.format) == fmt
        sa, sb = dict(a.named_parameters()), dict(b.named_parameters())
        assert sa.keys() == sb.keys() and sa["weight"].dtype == (torch.float8_e4m3fn if fp8 else torch.int8)
        for k in sa:
            assert sa[k].dtype == sb[k].dtype and sa[k].device == sb[k].device and sa[k].shape == sb[k].shape, k
            assert torch.equal(sa[k].view(torch.uint8) if sa[k].dtype == torch.float8_e4m3fn else sa[k], sb[k].view(torch.uint8) if fp8 and k == "weight" else sb[k]), k
    # against the oracle too (the per-module path is pinned elsewhere; this pins the batch directly)
    for m in ref:
        cta.decompress_module(m)
    comp.decompress_model(model)
    for a, b in zip(model, ref):
        assert eq(a.weight.data.cpu(), b.weight.data.cpu()) and a.weight.dtype == b.weight.dtype
        assert enum_name(a.quantization_status) == "decompressed"


def enum_name(v):
    return str(getattr(v, "value", v))


def test_q8_batch_vs_oracle(cta, dev):
    """the batched 8-bit C-ABI entry points against the CPU oracle: int8 (8 and 5 bits) and float8, bf16 and fp16, per-tensor /
    channel / group scales, with and without zero points, shapes that end inside a workgroup"""
    for dtype in (BF16, F16):
        for kind, bits in (("int8", 8), ("int8", 5), ("fp8", 8)):
            g = torch.Generator().manual_seed(11 + bits)
            entries, dent, refs = [], [], []
            for rows, cols, group, with_zp in ((40, 256, 128, True), (7, 64, 16, False), (300, 1024, 1024, True), (2, 16, 16, False), (513, 4096, 128, True),
                                               (64, 512, 64 * 512, False), (33, 48, 33 * 48, True)):
                w = torch.randn(rows, cols, generator=g).to(dtype)
                strategy = "tensor" if group == rows * cols else "channel" if group == cols else "group"
                if kind == "fp8":
                    s = O.calculate_qparams_float(w.reshape(1, -1) if strategy == "tensor" else w, kind="fp8", group_size=None if strategy != "group" else group)
                    scale, zp = (s.reshape(1) if strategy == "tensor" else s), None
                else:
                    scale, zp = O.calculate_qparams_minmax(w.reshape(1, -1) if strategy == "tensor" else w, num_bits=bits,
                                                           group_size=None if strategy != "group" else group, symmetric=not with_zp)
                    if strategy == "tensor":
                        scale, zp = scale.reshape(1), zp.reshape(1)
                    if not with_zp:
                        zp = None
                qdt = torch.float8_e4m3fn if kind == "fp8" else torch.int8
                q_ref = O.quantize(w, scale, zp, num_bits=bits, strategy=strategy, group_size=group if strategy == "group" else None, dtype=qdt,
                                   qtype="float" if kind == "fp8" else "int")
                d_ref = O.dequantize(q_ref, scale, zp)
                wd, sd_, zd = w.to(dev), scale.to(dev), d(zp, dev)
                assert cta.codec.q8_batch_group(w.shape, dtype, sd_, zd, device=dev, strategy=strategy, group_size=group) == group
                q = torch.empty((rows, cols), dtype=qdt, device=dev)
                out = torch.empty((rows, cols), dtype=dtype, device=dev)
                entries.append((wd, sd_, zd, q, rows, cols, group))
                dent.append((q, sd_, zd, out, rows, cols, group))
                refs.append((q_ref, d_ref))
            cta.codec.W4Batch(entries, "compress", dtype, kind=kind, bits=bits).launch()
            cta.codec.W4Batch(dent, "decompress", dtype, kind=kind).launch()
            for (wd, sd_, zd, q, rows, cols, group), (q2, sd2, zd2, out, *_), (q_ref, d_ref) in zip(entries, dent, refs):
                assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8)), (dtype, kind, bits, rows, cols, group)
                assert eq(out.cpu(), d_ref), (dtype, kind, bits, rows, cols, group)
    with pytest.raises(ValueError, match="not eligible"):
        x = torch.zeros(8, 24, dtype=BF16, device=dev)
        cta.codec.W4Batch([(x, x, None, x, 8, 24, 24)], "compress", BF16, kind="int8")


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("kind", ["fp8", "fp8z", "int8"])
def test_q8_batch_block_strategy_vs_oracle(cta, dev, dtype, kind):
    """block-strategy items (FP8-block checkpoints: scale[r // bh][c // bw], forward.py:198-216) in the 8-bit tables — group = -((bh << 24) | bw) — mixed with
    channel / group items in ONE launch, against the oracle: blocks 128 x 128, 64 x 128, 16 x 32, 1 x 16, row counts that are not a multiple of the block height,
    with zero points (float8 for FLOAT, int8 for INT) and without; then the class paths: compress_modules / decompress_modules of block-scheme modules through the C++ host
    loop against the per-module calls, and the layout the table refuses"""
    g = torch.Generator().manual_seed(5)
    qtype = "int" if kind == "int8" else "float"
    qdt = torch.int8 if kind == "int8" else F8
    entries, dent, refs = [], [], []
    cases = ((256, 512, (128, 128)), (200, 256, (64, 128)), (48, 96, (16, 32)), (5, 64, (1, 16)), (40, 256, None), (1024, 1024, (128, 128)), (130, 384, (128, 128)), (64, 512, 64))
    for rows, cols, blk in cases:
        w = (torch.randn(rows, cols, generator=g) * 2).to(dtype)
        if isinstance(blk, tuple):
            bh, bw = blk
            sshape = (-(-rows // bh), cols // bw)
            strategy, gs, block = "block", None, [bh, bw]
            group = -((bh << 24) | bw)
        elif blk is None:
            sshape, strategy, gs, block, group = (rows, 1), "channel", None, None, cols
        else:
            sshape, strategy, gs, block, group = (rows, cols // blk), "group", blk, None, blk
        scale = (torch.rand(sshape, generator=g) * 0.05 + 0.003).to(dtype)
        zp = None if kind == "fp8" else ((torch.randn(sshape, generator=g) * 3).to(F8) if kind == "fp8z" else torch.randint(-20, 20, sshape, generator=g, dtype=torch.int8))
        kw = dict(num_bits=8, strategy=strategy, group_size=gs, block_structure=block, qtype=qtype)
        q_ref = O.quantize(w, scale, zp, dtype=qdt, **kw)
        d_ref = O.dequantize(q_ref, scale, zp, strategy=strategy, group_size=gs, block_structure=block)
        wd, sd_, zd = w.to(dev), scale.to(dev), d(zp, dev)
        assert cta.codec.q8_batch_group(w.shape, dtype, sd_, zd, device=dev, strategy=strategy, group_size=gs, block_structure=block, f8_zero_point=kind == "fp8z") == group
        if strategy == "block" and rows % block[0] == 0 and block[0] > 1:  # the layout is also inferred from the scale's shape, as `dequantize` does (1-row blocks read as groups)
            assert cta.codec.q8_batch_group(w.shape, dtype, sd_, zd, device=dev, f8_zero_point=kind == "fp8z") == group
        q = torch.empty((rows, cols), dtype=qdt, device=dev)
        out = torch.empty((rows, cols), dtype=dtype, device=dev)
        entries.append((wd, sd_, zd, q, rows, cols, group))
        dent.append((q, sd_, zd, out, rows, cols, group))
        refs.append((q_ref, d_ref))
    cta.codec.W4Batch(entries, "compress", dtype, kind=kind, bits=8).launch()
    cta.codec.W4Batch(dent, "decompress", dtype, kind=kind).launch()
    for (wd, sd_, zd, q, rows, cols, group), (q2, sd2, zd2, out, *_), (q_ref, d_ref) in zip(entries, dent, refs):
        assert eq_f8(q.cpu(), q_ref) if qdt is F8 else torch.equal(q.cpu(), q_ref), (rows, cols, group)
        assert eq(out.cpu(), d_ref), (rows, cols, group)
    x = torch.zeros(96, 96, dtype=dtype, device=dev)
    with pytest.raises(ValueError, match="not eligible"):  # a block width that is not a power of two
        cta.codec.W4Batch([(x, x, None, x, 96, 96, -((32 << 24) | 48))], "compress", dtype, kind="int8")
    assert cta.codec.q8_batch_group((96, 96), dtype, torch.ones(3, 2, dtype=dtype, device=dev), None, device=dev, strategy="block", block_structure=[32, 48]) is None
    # the class paths
    wa = cta.QuantizationArgs(num_bits=8, type=qtype, strategy="block", block_structure=[128, 128], symmetric=kind != "int8")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa, input_activations=cta.QuantizationArgs(num_bits=8, type=qtype, strategy="tensor", symmetric=True, dynamic=True))
    klass = cta.BaseCompressor.get_value_from_registry("float-quantized" if qtype == "float" else "int-quantized")

    def modules():
        ms = []
        for k, (rows, cols) in enumerate(((256, 512), (1024, 1024), (130, 384), (128, 128))):
            gk = torch.Generator().manual_seed(100 + k)
            lin = torch.nn.Linear(cols, rows, bias=False, device="meta")
            lin.weight = torch.nn.Parameter((torch.randn(rows, cols, generator=gk) * 2).to(dtype).to(dev), requires_grad=False)
            sshape = (-(-rows // 128), cols // 128)
            lin.weight_scale = torch.nn.Parameter((torch.rand(sshape, generator=gk) * 0.05 + 0.003).to(dtype).to(dev), requires_grad=False)
            if kind != "fp8":
                zp = torch.zeros(sshape, dtype=F8) if kind == "fp8z" else torch.randint(-20, 20, sshape, generator=gk, dtype=torch.int8)
                lin.weight_zero_point = torch.nn.Parameter(zp.to(dev), requires_grad=False)
            lin.quantization_scheme = scheme
            ms.append(lin)
        return ms

    a, b = modules(), modules()
    for direction in ("compress", "decompress"):
        getattr(klass, direction + "_modules")(a)
        for m in b:
            getattr(klass, direction + "_module")(m)
        for x_, y_ in zip(a, b):
            assert list(x_._parameters) == list(y_._parameters)
            for name, tx in x_._parameters.items():
                ty = y_._parameters[name]
                if tx is None or ty is None:
                    assert tx is ty
                    continue
                assert tx.dtype == ty.dtype and (eq_f8(tx.data.cpu(), ty.data.cpu()) if tx.dtype is F8 else eq(tx.data.cpu(), ty.data.cpu())), (direction, name)


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("strategy", ["channel", "group", "tensor"])
def test_w8_pack_quantized_modules_through_the_tables(cta, dev, dtype, strategy):
    """pack-quantized with num_bits = 8 (the W8A16 preset): compress_modules / decompress_modules — the C++ host loop and the 8-bit tables' packed kind
    (int8 + 128, four codes to an int32 word) — against the per-module class calls (names, order, values) and against the oracle's pack_to_int32(quantize(...))
    and dequantize"""
    g = torch.Generator().manual_seed(21)
    wa = cta.QuantizationArgs(num_bits=8, type="int", strategy=strategy, group_size=128 if strategy == "group" else None, symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa)
    klass = cta.BaseCompressor.get_value_from_registry("pack-quantized")
    shapes = [(256, 512), (1024, 1024), (130, 384), (7, 256), (64, 4096)]
    ws = [(torch.randn(sh, generator=g) * 0.1).to(dtype) for sh in shapes]
    for w in ws:
        w.view(-1)[: special_values(dtype).numel()] = special_values(dtype)[: w.numel()]
    ws = [torch.where(torch.isfinite(w), w, torch.zeros_like(w)) for w in ws]
    qp = [O.calculate_qparams_minmax(w.reshape(1, -1) if strategy == "tensor" else w, num_bits=8, group_size=128 if strategy == "group" else None, symmetric=True) for w in ws]
    qp = [(s_.reshape(1), z_.reshape(1)) if strategy == "tensor" else (s_, z_) for s_, z_ in qp]

    def modules():
        ms = []
        for w, (s_, z_) in zip(ws, qp):
            lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, device="meta")
            lin.weight = torch.nn.Parameter(w.clone().to(dev), requires_grad=False)
            lin.weight_scale = torch.nn.Parameter(s_.clone().to(dev), requires_grad=False)
            lin.weight_zero_point = torch.nn.Parameter(z_.clone().to(dev), requires_grad=False)
            lin.quantization_scheme = scheme
            ms.append(lin)
        return ms

    a, b = modules(), modules()
    for direction in ("compress", "decompress"):
        getattr(klass, direction + "_modules")(a)
        for m in b:
            getattr(klass, direction + "_module")(m)
        for k, (x_, y_) in enumerate(zip(a, b)):
            assert list(x_._parameters) == list(y_._parameters), (direction, shapes[k])
            for name, tx in x_._parameters.items():
                ty = y_._parameters[name]
                if tx is None or ty is None:
                    assert tx is ty
                    continue
                assert tx.dtype == ty.dtype and eq(tx.data.cpu(), ty.data.cpu()), (direction, shapes[k], name)
            kw = dict(num_bits=8, strategy=strategy, group_size=128 if strategy == "group" else None)
            q_ref = O.quantize(ws[k], qp[k][0], qp[k][1], dtype=torch.int8, **kw)
            if direction == "compress":
                assert torch.equal(x_.weight_packed.cpu(), O.pack_to_int32(q_ref, 8).contiguous()) and x_.weight_shape.tolist() == list(shapes[k])
            else:
                assert eq(x_.weight.data.cpu(), O.dequantize(q_ref, qp[k][0], None, **{k_: v for k_, v in kw.items() if k_ != "num_bits"}))


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("with_zp", [False, True])
def test_mxfp8_modules_through_the_tables(cta, dev, dtype, with_zp):
    """mxfp8-quantized: compress_modules / decompress_modules — the C++ host loop, the 8-bit tables (float8 codes, groups of 32) and the scale table
    (ct_mx_scale_batch: E8M0 codes in, bfloat16 powers of two out) — against the per-module class calls: names, order, dtypes, values"""
    g = torch.Generator().manual_seed(31)
    wa = cta.QuantizationArgs(num_bits=8, type="float", strategy="group", group_size=32, symmetric=True, scale_dtype=torch.uint8)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa)
    scheme.format = "mxfp8-quantized"
    klass = cta.BaseCompressor.get_value_from_registry("mxfp8-quantized")
    shapes = [(256, 512), (1024, 1024), (130, 384), (7, 256), (64, 4096), (3, 64), (5, 96), (9, 160)]  # scale counts that end inside a lane's eight

    def modules():
        ms = []
        for k, (r, c) in enumerate(shapes):
            gk = torch.Generator().manual_seed(300 + k)
            w = (torch.randn(r, c, generator=gk) * 3).to(dtype)
            w.view(-1)[: min(special_values(dtype).numel(), w.numel())] = special_values(dtype)[: w.numel()]
            w = torch.where(torch.isfinite(w), w, torch.zeros_like(w))
            amax = w.float().reshape(r, -1, 32).abs().amax(-1).clamp(min=1e-4)
            s_ = torch.exp2(torch.floor(torch.log2(amax)) - 8).to(dtype)
            lin = torch.nn.Linear(c, r, bias=False, device="meta")
            lin.weight = torch.nn.Parameter(w.to(dev), requires_grad=False)
            lin.weight_scale = torch.nn.Parameter(s_.to(dev), requires_grad=False)
            if with_zp:
                lin.weight_zero_point = torch.nn.Parameter(torch.zeros(s_.shape, dtype=F8).to(dev), requires_grad=False)
            lin.quantization_scheme = scheme
            ms.append(lin)
        return ms

    a, b = modules(), modules()
    for direction in ("compress", "decompress"):
        getattr(klass, direction + "_modules")(a)
        for m in b:
            getattr(klass, direction + "_module")(m)
        for k, (x_, y_) in enumerate(zip(a, b)):
            assert list(x_._parameters) == list(y_._parameters), (direction, shapes[k])
            for name, tx in x_._parameters.items():
                ty = y_._parameters[name]
                if tx is None or ty is None:
                    assert tx is ty
                    continue
                assert tx.dtype == ty.dtype and (eq_f8(tx.data.cpu(), ty.data.cpu()) if tx.dtype is F8 else eq(tx.data.cpu(), ty.data.cpu())), (direction, shapes[k], name)
            if direction == "compress":
                assert x_.weight.dtype is F8 and x_.weight_scale.dtype is torch.uint8 and "weight_zero_point" not in x_._parameters
            else:
                assert x_.weight.dtype is BF16 and x_.weight_scale.dtype is BF16


@pytest.mark.parametrize("variant", ["w3", "w3_asym", "w2", "w6_channel", "w4_actorder", "w4_actorder_asym", "w8_asym", "w5_trainable_bias"])
def test_pack_quantized_modules_no_table_takes_match_the_per_module_calls(cta, dev, variant):
    """pack-quantized modules outside every table (2 / 3 / 5 / 6-bit words, activation ordering, asymmetric 8-bit): compress_modules / decompress_modules write the
    codec's result back as a delta (PackedQuantizationCompressor._delta_module) — the same names in the same order, the same kinds, dtypes and values as
    compress_module / decompress_module (replace_direct_state_dict) leave, and the oracle's words"""
    g = torch.Generator().manual_seed(41)
    bits = {"w3": 3, "w3_asym": 3, "w2": 2, "w6_channel": 6, "w4_actorder": 4, "w4_actorder_asym": 4, "w8_asym": 8, "w5_trainable_bias": 5}[variant]
    sym = variant not in ("w3_asym", "w8_asym", "w4_actorder_asym")
    strategy = "channel" if variant == "w6_channel" else "group"
    wa = cta.QuantizationArgs(num_bits=bits, type="int", strategy=strategy, group_size=None if strategy == "channel" else 128, symmetric=sym,
                              actorder="group" if variant.startswith("w4_actorder") else None)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa)
    klass = cta.BaseCompressor.get_value_from_registry("pack-quantized")
    shapes = [(256, 512), (130, 384), (64, 1024)]

    def modules():
        ms = []
        for k, (r, c) in enumerate(shapes):
            gk = torch.Generator().manual_seed(500 + k)
            w = (torch.randn(r, c, generator=gk) * 0.1).to(BF16)
            s_, z_ = O.calculate_qparams_minmax(w, num_bits=bits, group_size=None if strategy == "channel" else 128, symmetric=sym)
            lin = torch.nn.Linear(c, r, bias=variant == "w5_trainable_bias", device="meta")
            if variant == "w5_trainable_bias":
                lin.bias = torch.nn.Parameter(torch.zeros(r, dtype=BF16, device=dev), requires_grad=True)
            lin.weight = torch.nn.Parameter(w.to(dev), requires_grad=True)
            lin.weight_scale = torch.nn.Parameter(s_.to(dev), requires_grad=False)
            lin.weight_zero_point = torch.nn.Parameter(z_.to(dev), requires_grad=False)
            if variant.startswith("w4_actorder"):
                perm = torch.randperm(c, generator=gk)
                lin.weight_g_idx = torch.nn.Parameter((perm // 128).to(torch.int32).to(dev), requires_grad=False)
            lin.quantization_scheme = scheme
            ms.append(lin)
        return ms

    a, b = modules(), modules()
    for direction in ("compress", "decompress"):
        getattr(klass, direction + "_modules")(a)
        for m in b:
            getattr(klass, direction + "_module")(m)
        for k, (x_, y_) in enumerate(zip(a, b)):
            assert list(x_._parameters) == list(y_._parameters) and list(x_._buffers) == list(y_._buffers), (direction, shapes[k], list(x_._parameters), list(y_._parameters))
            assert x_.quantization_status == y_.quantization_status
            for name, tx in x_._parameters.items():
                ty = y_._parameters[name]
                if tx is None or ty is None:
                    assert tx is ty
                    continue
                assert type(tx) is type(ty) and tx.requires_grad == ty.requires_grad and tx.dtype == ty.dtype and tx.device == ty.device, (direction, name)
                assert eq(tx.data.cpu(), ty.data.cpu()) if tx.dtype.is_floating_point else torch.equal(tx.data.cpu(), ty.data.cpu()), (direction, shapes[k], name)


def test_gidx_col_group_on_the_device(cta, dev):
    """ct_gidx_col_group (the activation-ordering group table: rank in the stable sort of g_idx // group_size, or plain column order while a -1 is left) against
    the reference's expression (forward_helpers.py:147-175) evaluated on the CPU: balanced permutations (no sort needed), unbalanced and out-of-range values,
    more groups than the histogram has bins, a -1 anywhere, one column"""
    from compressed_tensors_amd.codec import _col_group_of

    def ref(flat, gs):
        inv = torch.argsort(torch.argsort(flat, stable=True), stable=True)
        plain = torch.arange(flat.numel())
        return (torch.where((flat == -1).any(), plain, inv) // gs).to(torch.int32)

    g = torch.Generator().manual_seed(61)
    cases = []
    for cols, gs in ((256, 128), (4096, 128), (28672, 128), (14336, 32), (384, 128), (1024, 1024), (96, 32)):
        cases.append((torch.randperm(cols, generator=g) // gs, gs, "balanced"))
    for cols, gs in ((4096, 128), (1000, 128), (5000, 64), (257, 16)):
        cases.append((torch.randint(0, max(cols // gs, 1) + 3, (cols,), generator=g), gs, "unbalanced"))
    cases.append((torch.randint(-5, 100000, (3000,), generator=g), 128, "any values"))
    cases.append((torch.randperm(8192, generator=g), 1, "more groups than bins"))
    minus = torch.randperm(4096, generator=g) // 128
    minus[777] = -1
    cases.append((minus, 128, "a -1 left"))
    cases.append((torch.full((512,), -1), 128, "not initialised"))
    cases.append((torch.zeros(1, dtype=torch.int64), 128, "one column"))
    for flat, gs, what in cases:
        flat = flat.to(torch.int32)
        got = _col_group_of(flat.to(dev), gs)
        assert got.dtype is torch.int32 and got.data_ptr() % 16 == 0 and torch.equal(got.cpu(), ref(flat.long(), gs)), (what, flat.numel(), gs)


@pytest.mark.parametrize("fmt,group", [("nvfp4-pack-quantized", 16), ("mxfp4-pack-quantized", 32)])
def test_fp4_decompress_many_equals_decompress(cta, dev, fmt, group):
    """NVFP4 / MXFP4 `decompress_many` (the model-free converter's call: a shard's state dicts through ONE table launch) against `decompress` per state dict:
    same keys in the same order, same dtypes and bytes; a state dict outside the table's layout (CPU tensors) takes the single path inside the same call"""
    klass = cta.BaseCompressor.get_value_from_registry(fmt)
    scheme = _fp4_scheme(cta, fmt)
    g = torch.Generator().manual_seed(81)
    sds = []
    for k, (r, c) in enumerate(((256, 512), (7, 64), (1024, 1024), (33, 1056), (64, 128))):
        x = (torch.randn(r, c, generator=g) * (1 + k)).to(BF16)
        if group == 16:
            gs = O.generate_gparam(x)
            s_ = O.calculate_qparams_float(x, kind="nvfp4", group_size=16, global_scale=gs)
        else:
            gs, s_ = None, O.calculate_qparams_float(x, kind="mxfp4", group_size=32)
        sd = {"weight": x.to(dev), "weight_scale": s_.to(dev)}
        if gs is not None:
            sd["weight_global_scale"] = gs.to(dev)
        c_ = klass.compress(sd, scheme)
        if k == 4:
            c_ = {kk: v.cpu() for kk, v in c_.items()}  # not the table's layout: decompressed by the single path (staged through the GPU)
        sds.append(c_)
    many = klass.decompress_many(sds, scheme)
    for sd, got in zip(sds, many):
        ref = klass.decompress(sd, scheme)
        assert list(got) == list(ref)
        for kk in ref:
            assert got[kk].dtype == ref[kk].dtype and got[kk].shape == ref[kk].shape and got[kk].device == ref[kk].device, kk
            assert torch.equal(got[kk].contiguous().view(torch.uint8).cpu(), ref[kk].contiguous().view(torch.uint8).cpu()), kk


def test_mxfp8_decompress_many_equals_decompress(cta, dev):
    """mxfp8-quantized `decompress_many` (a shard's state dicts: one scale-table launch + one launch of the 8-bit tables) against `decompress` per state dict: same
    keys in the same order, dtypes and bytes; a CPU state dict takes the single path inside the same call"""
    klass = cta.BaseCompressor.get_value_from_registry("mxfp8-quantized")
    wa = cta.QuantizationArgs(num_bits=8, type="float", strategy="group", group_size=32, symmetric=True, scale_dtype=torch.uint8)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa)
    scheme.format = "mxfp8-quantized"
    g = torch.Generator().manual_seed(91)
    sds = []
    for k, (r, c) in enumerate(((256, 512), (7, 64), (1024, 1024), (33, 1056), (64, 128))):
        x = (torch.randn(r, c, generator=g) * (1 + k)).to(BF16)
        s_ = torch.exp2(torch.floor(torch.log2(x.float().reshape(r, -1, 32).abs().amax(-1).clamp(min=1e-4))) - 8).to(BF16)
        c_ = klass.compress({"weight": x.to(dev), "weight_scale": s_.to(dev)}, scheme)
        if k == 4:
            c_ = {kk: v.cpu() for kk, v in c_.items()}
        sds.append(c_)
    many = klass.decompress_many(sds, scheme)
    for sd, got in zip(sds, many):
        ref = klass.decompress(sd, scheme)
        assert list(got) == list(ref)
        for kk in ref:
            assert got[kk].dtype == ref[kk].dtype and got[kk].shape == ref[kk].shape and got[kk].device == ref[kk].device, kk
            assert torch.equal(got[kk].contiguous().view(torch.uint8).cpu(), ref[kk].contiguous().view(torch.uint8).cpu()), kk


def test_model_compressor_on_a_tree_of_mixed_formats(cta, dev):
    """one model whose modules carry different schemes — W4 g128, W4 asymmetric, activation-ordered W4, W3, W8A16, FP8 channel, FP8 block, MXFP8, NVFP4, MXFP4 —
    through ModelCompressor.compress_model / decompress_model (every format's own C++ loop and table in one call) against compress_module / decompress_module
    on a twin of each module: names, order, dtypes, bytes"""
    import copy

    from compressed_tensors_amd.compressors.base import compress_module, decompress_module

    QA, QS = cta.QuantizationArgs, cta.QuantizationScheme
    g = torch.Generator().manual_seed(71)
    schemes = {
        "w4": QS(targets=["Linear"], weights=QA(num_bits=4, group_size=128, symmetric=True, strategy="group")),
        "w4asym": QS(targets=["Linear"], weights=QA(num_bits=4, group_size=128, symmetric=False, strategy="group")),
        "w4act": QS(targets=["Linear"], weights=QA(num_bits=4, group_size=128, symmetric=True, strategy="group", actorder="group")),
        "w3": QS(targets=["Linear"], weights=QA(num_bits=3, group_size=128, symmetric=True, strategy="group")),
        "w8a16": QS(targets=["Linear"], weights=QA(num_bits=8, symmetric=True, strategy="channel")),
        "fp8": QS(targets=["Linear"], weights=QA(num_bits=8, type="float", strategy="channel", symmetric=True),
                  input_activations=QA(num_bits=8, type="float", strategy="tensor", symmetric=True, dynamic=True)),
        "fp8blk": QS(targets=["Linear"], weights=QA(num_bits=8, type="float", strategy="block", block_structure=[128, 128], symmetric=True),
                     input_activations=QA(num_bits=8, type="float", strategy="tensor", symmetric=True, dynamic=True)),
        "mxfp8": QS(targets=["Linear"], weights=QA(num_bits=8, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=torch.uint8)),
        "nvfp4": QS(targets=["Linear"], weights=QA(num_bits=4, type="float", strategy="tensor_group", symmetric=True, group_size=16, scale_dtype=F8)),
        "mxfp4": QS(targets=["Linear"], weights=QA(num_bits=4, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=torch.uint8)),
    }
    schemes["mxfp8"].format = "mxfp8-quantized"
    schemes["nvfp4"].format = "nvfp4-pack-quantized"
    schemes["mxfp4"].format = "mxfp4-pack-quantized"
    root = torch.nn.Module()
    root.layers = torch.nn.ModuleList()
    twins = []
    for rep in range(2):
        for name, scheme in schemes.items():
            r, c = (256, 512) if rep == 0 else (384, 1024)
            w = (torch.randn(r, c, generator=g) * 0.3).to(BF16)
            lin = torch.nn.Linear(c, r, bias=False, device="meta")
            lin.weight = torch.nn.Parameter(w.to(dev), requires_grad=False)
            wa = scheme.weights
            entries = {}
            if name in ("fp8",):
                entries["weight_scale"] = O.calculate_qparams_float(w, kind="fp8")
                entries["weight_zero_point"] = torch.zeros(r, 1, dtype=F8)
            elif name == "fp8blk":
                entries["weight_scale"] = (w.float().reshape(r // 128, 128, c // 128, 128).abs().amax(dim=(1, 3)) / 448.0).to(BF16)
            elif name == "mxfp8":
                entries["weight_scale"] = torch.exp2(torch.floor(torch.log2(w.float().reshape(r, -1, 32).abs().amax(-1).clamp(min=1e-4))) - 8).to(BF16)
            elif name == "nvfp4":
                gs_ = O.generate_gparam(w)
                entries["weight_global_scale"] = gs_
                entries["weight_scale"] = O.calculate_qparams_float(w, kind="nvfp4", group_size=16, global_scale=gs_)
            elif name == "mxfp4":
                entries["weight_scale"] = O.calculate_qparams_float(w, kind="mxfp4", group_size=32)
            else:
                s_, z_ = O.calculate_qparams_minmax(w, num_bits=int(wa.num_bits), group_size=getattr(wa, "group_size", None), symmetric=bool(wa.symmetric))
                entries["weight_scale"], entries["weight_zero_point"] = s_, z_
                if name == "w4act":
                    entries["weight_g_idx"] = (torch.randperm(c, generator=g) // 128).to(torch.int32)
            for k, v in entries.items():
                setattr(lin, k, torch.nn.Parameter(v.to(dev), requires_grad=False))
            lin.quantization_scheme = scheme
            root.layers.append(lin)
            twin = copy.deepcopy(lin)
            twin.quantization_scheme = scheme
            twins.append((name, twin))
    mc = cta.ModelCompressor()
    for direction, per_module in (("compress", compress_module), ("decompress", decompress_module)):
        getattr(mc, direction + "_model")(root)
        for lin, (name, twin) in zip(root.layers, twins):
            per_module(twin)
            assert list(lin._parameters) == list(twin._parameters), (direction, name, list(lin._parameters), list(twin._parameters))
            for k, t in lin._parameters.items():
                u = twin._parameters[k]
                if t is None or u is None:
                    assert t is u
                    continue
                assert t.dtype == u.dtype and t.shape == u.shape and t.device == u.device and torch.equal(t.data.contiguous().view(torch.uint8).cpu(), u.data.contiguous().view(torch.uint8).cpu()), (direction, name, k)
            assert lin.quantization_status == twin.quantization_status


def test_w4_batch_vs_oracle(cta, dev):
    """the batched C-ABI entry points against the CPU oracle, bf16 and fp16, group and channel"""
    for dtype in (BF16, F16):
        g = torch.Generator().manual_seed(9)
        items, entries, outs = [], [], []
        for rows, cols, group in ((40, 256, 128), (7, 64, 32), (300, 1024, 1024), (1, 32, 32), (513, 4096, 128)):
            w = torch.randn(rows, cols, generator=g).to(dtype)
            strategy = "channel" if group == cols else "group"
            scale, zp = O.calculate_qparams_minmax(w, num_bits=4, group_size=None if strategy == "channel" else group, symmetric=False)
            q = O.quantize(w, scale, zp, num_bits=4, strategy=strategy, group_size=group, dtype=torch.int8)
            items.append((w, scale, zp, O.pack_to_int32(q, 4), strategy, group))
            wd, sd, zd = w.to(dev), scale.to(dev), zp.to(dev)
            packed = torch.empty(rows, cols // 8, dtype=torch.int32, device=dev)
            entries.append((wd, sd, zd, packed, rows, cols, group))
        cta.codec.W4Batch(entries, "compress", dtype).launch()
        dent = []
        for (w, scale, zp, ref_packed, strategy, group), e in zip(items, entries):
            assert torch.equal(e[3].cpu(), ref_packed.contiguous())
            out = torch.empty_like(e[0])
            dent.append((e[3], e[1], e[2], out, e[4], e[5], e[6]))
        cta.codec.W4Batch(dent, "decompress", dtype).launch()
        for (w, scale, zp, ref_packed, strategy, group), e in zip(items, dent):
            q = O.unpack_from_int32(ref_packed, 4, w.shape)
            assert eq(e[3].cpu(), O.dequantize(q, scale, zp))


# ----------------------------------------------------------------------------- FP4 codecs
def _fp4_scheme(cta, fmt):
    from compressed_tensors_amd.quantization import QuantizationArgs, QuantizationScheme

    if fmt == "nvfp4-pack-quantized":
        w = QuantizationArgs(num_bits=4, type="float", strategy="tensor_group", symmetric=True, group_size=16, scale_dtype=torch.float8_e4m3fn)
    else:
        w = QuantizationArgs(num_bits=4, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=torch.uint8)
    return QuantizationScheme(targets=["Linear"], weights=w)


@pytest.mark.parametrize("case", cases("fp4"), ids=lambda c: c["key"])
def test_fp4_compressor_golden(golden, cta, dev, case):
    """the reference's own outputs (tests/golden/fp4.safetensors, oracle/gen_golden.py) through the compressor classes"""
    t = golden.case("fp4", case["key"])
    fmt = case["format"]
    comp = cta.BaseCompressor.get_value_from_registry(fmt)
    scheme = _fp4_scheme(cta, fmt)
    sd = {k[3:]: d(v, dev) for k, v in t.items() if k.startswith("in.")}
    c = comp.compress(sd, scheme)
    assert sorted(c) == case["compressed_keys"]
    assert c["weight_packed"].is_cuda and c["weight_packed"].dtype == torch.uint8
    assert torch.equal(c["weight_packed"].cpu(), t["comp.weight_packed"])
    assert str(c["weight_scale"].dtype) == "torch." + case["compressed_scale_dtype"]
    assert torch.equal(c["weight_scale"].cpu().view(torch.uint8), t["comp.weight_scale"])
    back = comp.decompress(c, scheme)
    assert sorted(back) == case["decompressed_keys"]
    for name in ("weight", "weight_scale"):
        assert eq(back[name].cpu(), t[f"dec.{name}"]), name


def _all_16bit(dtype):
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dtype)
    return bits[~torch.isnan(bits)]


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_fp4_hardware_conversion_all_inputs(cta, dev, xdt):
    """v_cvt_scalef32_pk_fp4_f32 against cast_to_fp4's thresholds: EVERY non-NaN 16-bit input (ties, +-0, subnormals,
    +-inf, saturation), through both the in-dtype (mxfp4) and the float32 (nvfp4) quotient paths, at several scales"""
    x = _all_16bit(xdt)
    n = (x.numel() // 32) * 32
    x = x[torch.randperm(x.numel(), generator=torch.Generator().manual_seed(0))][:n].reshape(-1, 32).contiguous()
    rows = x.shape[0]
    for sval in (1.0, 0.25, 2.0 ** -9, 64.0):  # mxfp4: power-of-two scales in x's dtype
        s = torch.full((rows, 1), sval, dtype=xdt)
        got = cta.codec.fp4_quantize_and_pack(d(x, dev), d(s, dev), None, group_size=32)
        ref = O.fp4_compress(x, s, None, fmt="mxfp4-pack-quantized")["weight_packed"]
        assert torch.equal(got.cpu(), ref), sval
    g = torch.Generator().manual_seed(1)
    for gsval in (1.0, 2688.0 / 7.3, 0.37):  # nvfp4: fp8-representable float32 scales under a float32 global scale
        s = (torch.rand((rows, 2), generator=g) * 400 + 0.002).to(torch.float8_e4m3fn).to(torch.float32)
        gs = torch.tensor([gsval], dtype=torch.float32)
        got = cta.codec.fp4_quantize_and_pack(d(x, dev), d(s, dev), d(gs, dev), group_size=16)
        ref = O.fp4_compress(x, s, gs, fmt="nvfp4-pack-quantized")["weight_packed"]
        assert torch.equal(got.cpu(), ref), gsval


def test_fp4_decode_all_codes_all_scales(cta, dev):
    """every packed byte under every fp8-e4m3 scale (nvfp4) and every E8M0 exponent (mxfp4)"""
    codes = torch.arange(256, dtype=torch.uint8)
    # nvfp4: row r uses fp8 scale byte r for its single group of 16 (8 bytes); 32 rows of byte patterns per scale
    sbytes = torch.arange(256, dtype=torch.uint8)
    sbytes = sbytes[~torch.isnan(sbytes.view(torch.float8_e4m3fn).float())]
    packed = codes.reshape(32, 8).repeat(sbytes.numel(), 1).contiguous()
    scale = sbytes.repeat_interleave(32).reshape(-1, 1).view(torch.float8_e4m3fn)
    for gsval in (1.0, 0.37, 2688.0 / 7.3):
        gs = torch.tensor([gsval], dtype=torch.float32)
        got = cta.codec.fp4_unpack_and_dequantize(d(packed, dev), d(scale, dev), d(gs, dev), group_size=16, scale_kind="f8e4m3")
        ref = O.fp4_decompress({"weight_packed": packed, "weight_scale": scale, "weight_global_scale": gs}, fmt="nvfp4-pack-quantized")["weight"]
        assert got.dtype == BF16 and eq(got.cpu(), ref), gsval
    e = torch.arange(256, dtype=torch.uint8)
    packed = codes.reshape(16, 16).repeat(256, 1).contiguous()
    scale = e.repeat_interleave(16).reshape(-1, 1)
    got = cta.codec.fp4_unpack_and_dequantize(d(packed, dev), d(scale, dev), None, group_size=32, scale_kind="e8m0")
    ref = O.fp4_decompress({"weight_packed": packed, "weight_scale": scale}, fmt="mxfp4-pack-quantized")["weight"]
    assert eq(got.cpu(), ref)


@pytest.mark.parametrize("fmt,group", [("nvfp4-pack-quantized", 16), ("mxfp4-pack-quantized", 32)])
@pytest.mark.parametrize("xdt", [BF16, F16])
@pytest.mark.parametrize("shape", [(1, 32), (3, 96), (5, 160), (64, 4096), (33, 1056), (1, 16), (3, 48), (7, 80)])
def test_fp4_codec_vs_oracle(cta, dev, fmt, group, xdt, shape):
    if shape[1] % group:
        pytest.skip("columns not a multiple of the group")
    g = torch.Generator().manual_seed(shape[0] * 131 + shape[1])
    x = (torch.randn(shape, generator=g) * 3).to(xdt)
    x.view(-1)[: special_values(xdt).numel()] = special_values(xdt)[: x.numel()]
    x = torch.where(torch.isnan(x), torch.zeros_like(x), x)
    amax = x.float().reshape(shape[0], -1, group).abs().amax(-1).clamp(min=1e-3, max=1e4)
    if fmt.startswith("nvfp4"):
        gs = torch.tensor([448.0 * 6.0 / float(amax.max())], dtype=torch.float32)
        s = (gs * amax / 6.0).to(torch.float8_e4m3fn).to(torch.float32)
        s = torch.where(s == 0, torch.full_like(s, 2.0 ** -9), s)
    else:
        gs = None
        s = torch.exp2(torch.floor(torch.log2(amax)) - 2).to(xdt)
    comp = cta.BaseCompressor.get_value_from_registry(fmt)
    scheme = _fp4_scheme(cta, fmt)
    sd = {"weight": d(x, dev), "weight_scale": d(s, dev)}
    if gs is not None:
        sd["weight_global_scale"] = d(gs, dev)
    c = comp.compress(sd, scheme)
    ref = O.fp4_compress(x, s, gs, fmt=fmt)
    assert torch.equal(c["weight_packed"].cpu(), ref["weight_packed"])
    assert torch.equal(c["weight_scale"].cpu().view(torch.uint8), ref["weight_scale"].view(torch.uint8))
    back = comp.decompress(c, scheme)
    rback = O.fp4_decompress(ref, fmt=fmt)
    assert back["weight"].dtype == BF16 and eq(back["weight"].cpu(), rback["weight"])
    assert eq(back["weight_scale"].cpu(), rback["weight_scale"])


@pytest.mark.parametrize("fmt,group", [("nvfp4-pack-quantized", 16), ("mxfp4-pack-quantized", 32)])
@pytest.mark.parametrize("xdt,sdt", [(BF16, BF16), (F16, F16), (BF16, F32)])
def test_fp4_table_launches_vs_oracle(cta, dev, fmt, group, xdt, sdt):
    """ct_fp4_batch_plan + ct_fp4_quant_pack_batch / ct_fp4_unpack_dequant_batch (one launch for a table of tensors, each with its own global scale) against
    the oracle per tensor: packed nibbles, stored scales (float8 bytes / E8M0 codes), dense bfloat16 weights and bfloat16 scales; shapes from one lane to
    several workgroups, incl. sizes that end inside a workgroup's first / second chunk, special values; then the same modules through
    compress_modules / decompress_modules (the C++ host loop + the table launch), state dicts equal to the per-module class calls"""
    from compressed_tensors_amd import codec

    if group == 32 and sdt is F32:
        pytest.skip("MXFP4 scales are 16-bit")
    shapes = [(1, 32), (2, 16), (3, 96), (64, 4096), (33, 1056), (7, 160), (256, 2048), (1024, 1024), (129, 992), (16, 4128)]
    shapes = [sh for sh in shapes if sh[1] % group == 0 and (sh[0] * sh[1]) % 32 == 0]
    g = torch.Generator().manual_seed(11)
    xs, ss, gss, refs = [], [], [], []
    for k, shape in enumerate(shapes):
        x = (torch.randn(shape, generator=g) * (0.5 + k)).to(xdt)
        sv = special_values(xdt)
        x.view(-1)[: min(sv.numel(), x.numel())] = sv[: x.numel()]
        x = torch.where(torch.isnan(x), torch.zeros_like(x), x)
        amax = x.float().reshape(shape[0], -1, group).abs().amax(-1).clamp(min=1e-3, max=1e4)
        if group == 16:
            gs = torch.tensor([448.0 * 6.0 / float(amax.max())], dtype=torch.float32)
            s = (gs * amax / 6.0).to(F8).to(torch.float32)
            s = torch.where(s == 0, torch.full_like(s, 2.0 ** -9), s).to(sdt)
        else:
            gs = None
            s = torch.exp2(torch.floor(torch.log2(amax)) - 2).to(sdt)
        xs.append(x); ss.append(s); gss.append(gs)
        refs.append(O.fp4_compress(x, s, gs, fmt=fmt))
    import array

    IW = codec._ITEM_WORDS
    dx, ds, dg = [t.to(dev) for t in xs], [t.to(dev) for t in ss], [None if t is None else t.to(dev) for t in gss]
    packed = [torch.empty((sh[0], sh[1] // 2), dtype=torch.uint8, device=dev) for sh in shapes]
    stored = [torch.empty((sh[0], sh[1] // group), dtype=F8 if group == 16 else torch.uint8, device=dev) for sh in shapes]
    flat = []
    for x, s_, g_, p_, st, sh in zip(dx, ds, dg, packed, stored, shapes):
        flat += [x.data_ptr(), s_.data_ptr(), 0 if g_ is None else g_.data_ptr(), p_.data_ptr(), sh[0], sh[1], group, 0, 0, 0, st.data_ptr()] + [0] * (IW - 11)
    words = torch.tensor(flat, dtype=torch.int64)
    codec.launch_fp4_words(words, len(shapes), "compress", dev, group, xdt, sdt)
    for k, ref in enumerate(refs):
        assert torch.equal(packed[k].cpu(), ref["weight_packed"]), (shapes[k], "packed")
        assert torch.equal(stored[k].cpu().view(torch.uint8), ref["weight_scale"].view(torch.uint8)), (shapes[k], "stored scale")
    outs = [torch.empty(sh, dtype=BF16, device=dev) for sh in shapes]
    souts = [torch.empty((sh[0], sh[1] // group), dtype=BF16, device=dev) for sh in shapes]
    flat = []
    for p_, st, g_, o_, so, sh in zip(packed, stored, dg, outs, souts, shapes):
        flat += [p_.data_ptr(), st.data_ptr(), 0 if g_ is None else g_.data_ptr(), o_.data_ptr(), sh[0], sh[1], group, 0, 0, 0, so.data_ptr()] + [0] * (IW - 11)
    codec.launch_fp4_words(torch.tensor(flat, dtype=torch.int64), len(shapes), "decompress", dev, group)
    for k, ref in enumerate(refs):
        state = dict(ref)
        if gss[k] is not None:
            state["weight_global_scale"] = gss[k]
        rb = O.fp4_decompress(state, fmt=fmt)
        assert eq(outs[k].cpu(), rb["weight"]), (shapes[k], "weight")
        assert eq(souts[k].cpu(), rb["weight_scale"]), (shapes[k], "bf16 scale")
    # a table that holds an item outside the layout is refused with nothing launched
    bad = words.clone()
    bad[5] = 24  # cols not a multiple of the group
    with pytest.raises(ValueError):
        codec.launch_fp4_words(bad, len(shapes), "compress", dev, group, xdt, sdt)
    # the module loops: C++ host loop + table launches against the per-module class calls
    comp = cta.BaseCompressor.get_value_from_registry(fmt)
    scheme = _fp4_scheme(cta, fmt)

    def modules():
        ms = []
        for x, s_, g_ in zip(dx, ds, dg):
            lin = torch.nn.Linear(x.shape[1], x.shape[0], bias=False, device="meta")
            lin.weight = torch.nn.Parameter(x.clone(), requires_grad=False)
            lin.weight_scale = torch.nn.Parameter(s_.clone(), requires_grad=False)
            if g_ is not None:
                lin.weight_global_scale = torch.nn.Parameter(g_.clone(), requires_grad=False)
            lin.quantization_scheme = scheme
            ms.append(lin)
        return ms

    a, b = modules(), modules()
    comp.compress_modules(a)
    for m in b:
        comp.compress_module(m)
    for k, (x_, y_) in enumerate(zip(a, b)):
        assert list(x_._parameters) == list(y_._parameters), shapes[k]
        for name in x_._parameters:
            tx, ty = x_._parameters[name], y_._parameters[name]
            if tx is None or ty is None:
                assert tx is ty
                continue
            assert tx.dtype == ty.dtype and torch.equal(tx.view(torch.uint8) if tx.element_size() == 1 else tx, ty.view(torch.uint8) if ty.element_size() == 1 else ty), (shapes[k], name)
        assert torch.equal(x_.weight_packed.cpu(), refs[k]["weight_packed"])
    comp.decompress_modules(a)
    for m in b:
        comp.decompress_module(m)
    for k, (x_, y_) in enumerate(zip(a, b)):
        assert list(x_._parameters) == list(y_._parameters), shapes[k]
        for name in x_._parameters:
            if x_._parameters[name] is None or y_._parameters[name] is None:
                assert x_._parameters[name] is y_._parameters[name]
                continue
            assert eq(x_._parameters[name].data.cpu(), y_._parameters[name].data.cpu()), (shapes[k], name)


def test_fp4_full_size_roundtrip(cta, dev):
    """8192 x 8192: decompress(compress(x)) re-compresses to the same bytes (idempotence), and E2M1-valued inputs
    under unit scales survive exactly"""
    N = 8192
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((N, N), generator=g, device=dev, dtype=BF16)
    s = torch.full((N, N // 32), 0.5, dtype=BF16, device=dev)
    p = cta.codec.fp4_quantize_and_pack(x, s, None, group_size=32)
    y = cta.codec.fp4_unpack_and_dequantize(p, s, None, group_size=32)
    p2 = cta.codec.fp4_quantize_and_pack(y, s, None, group_size=32)
    # a negative value that rounded to -0.0 comes back as -0.0, which compresses to +0 (sign(-0.0) == 0 upstream)
    neg_zero = ((p & 0x0F) == 0x08) | ((p & 0xF0) == 0x80)
    fixed = torch.where((p & 0x0F) == 0x08, p & 0xF0, p)
    fixed = torch.where((fixed & 0xF0) == 0x80, fixed & 0x0F, fixed)
    assert torch.equal(p2, fixed) and bool(neg_zero.any())
    grid = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0], device=dev)
    v = grid[torch.randint(0, grid.numel(), (N, N), generator=g, device=dev)].to(BF16)
    one = torch.ones((N, N // 32), dtype=BF16, device=dev)
    assert torch.equal(cta.codec.fp4_unpack_and_dequantize(cta.codec.fp4_quantize_and_pack(v, one, None, group_size=32), one, None, group_size=32), v)


def test_fp4_format_inference_and_errors(cta, dev):
    from compressed_tensors_amd.compressors.format import infer_module_format

    assert infer_module_format(torch.nn.Linear, _fp4_scheme(cta, "nvfp4-pack-quantized")).value == "nvfp4-pack-quantized"
    assert infer_module_format(torch.nn.Linear, _fp4_scheme(cta, "mxfp4-pack-quantized")).value == "mxfp4-pack-quantized"
    with pytest.raises(ValueError):
        cta.codec.fp4_quantize_and_pack(torch.zeros((2, 31), dtype=BF16, device=dev), torch.ones((2, 1), dtype=BF16, device=dev), None, group_size=32)


def test_fp4_primitives_golden(golden, cta, dev):
    """cast_to_fp4 / pack_fp4_to_uint8 / unpack_fp4_from_uint8 / E8M0 scales against the reference's own outputs"""
    t = golden.tensors("fp4")
    for dt in ("torch.float32", "torch.bfloat16", "torch.float16"):
        x, ref = t[f"cast_{dt}.in"], t[f"cast_{dt}.out"]
        got = cta.codec.cast_to_fp4(d(x, dev)).cpu()
        assert got.dtype == ref.dtype and torch.equal(got, ref) and torch.equal(torch.signbit(got), torch.signbit(ref))
    assert torch.equal(cta.codec.pack_fp4_to_uint8(d(t["pack.in"], dev)).cpu(), t["pack.out"])
    m, n = t["unpack.out"].shape
    un = cta.codec.unpack_fp4_from_uint8(d(t["pack.out"], dev), m, n, dtype=t["unpack.out"].dtype).cpu()
    assert torch.equal(un, t["unpack.out"]) and torch.equal(torch.signbit(un), torch.signbit(t["unpack.out"]))
    assert torch.equal(cta.codec.compress_mx_scale(d(t["e8m0.in"], dev)).cpu(), t["e8m0.out"])
    assert torch.equal(cta.codec.decompress_mx_scale(d(t["e8m0.out"], dev)).cpu(), t["e8m0.back"])


@pytest.mark.parametrize("dtype", [BF16, F16, F32])
def test_fp4_primitives_vs_oracle(cta, dev, dtype):
    """every 16-bit pattern (NaNs included for the cast), float32 neighbours of every rounding threshold, ragged sizes"""
    if dtype == F32:
        th = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0, 0.0, 1e-45, 3e38], dtype=F32)
        x = torch.cat([torch.nextafter(th, torch.full_like(th, 10.0)), th, torch.nextafter(th, torch.full_like(th, -10.0))])
        x = torch.cat([x, -x, torch.tensor([float("inf"), -float("inf"), float("nan"), -0.0]), torch.randn(4097) * 3])
    else:
        x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dtype)
    for n in (x.numel(), 1, 7, 9, 1001):
        xs = x[:n].contiguous()
        got = cta.codec.cast_to_fp4(d(xs, dev)).cpu()
        ref = O.cast_to_fp4(xs)
        nan = torch.isnan(xs)
        assert torch.equal(torch.isnan(got), nan)
        assert torch.equal(got[~nan], ref[~nan]) and torch.equal(torch.signbit(got[~nan]), torch.signbit(ref[~nan]))
    vals = O.cast_to_fp4(x[~torch.isnan(x)])
    for cols in (2, 6, 8, 30, 1024):
        v = vals[: (vals.numel() // cols) * cols].reshape(-1, cols).contiguous()
        packed = cta.codec.pack_fp4_to_uint8(d(v, dev))
        assert torch.equal(packed.cpu(), O.pack_fp4(O.fp4_nibbles_of_values(v)))
        for odt in (BF16, F16, F32):
            un = cta.codec.unpack_fp4_from_uint8(packed, v.shape[0], cols, dtype=odt).cpu()
            assert un.dtype == odt and torch.equal(un, v.to(odt)) and torch.equal(torch.signbit(un), torch.signbit(v))
    with pytest.raises(ValueError):
        cta.codec.pack_fp4_to_uint8(torch.zeros((2, 3), dtype=dtype, device=dev))


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_fp4_reciprocal_quotient_is_exact(cta, dev, xdt):
    """all 2^23 scale mantissas x all weight mantissas: the shared-reciprocal quotient == the IEEE fp32 divide"""
    assert cta.codec.selftest_fp4_div(xdt) == 0


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_fp4_quotient_range_guards(cta, dev, xdt):
    """effective scales on both sides of the fast-path ranges, tiny / huge / subnormal weights, non-power-of-two MX
    scales: every branch of the quotient against the oracle"""
    x = _all_16bit(xdt)
    n = (x.numel() // 32) * 32
    x = x[:n].reshape(-1, 32).contiguous()
    rows = x.shape[0]
    g = torch.Generator().manual_seed(5)
    for gsval in (2.0 ** -30, 2.0 ** -19.5, 2.0 ** -9, 1.0, 2.0 ** 12, 2.0 ** 21, 3e30):  # s_eff = s / gs sweeps 2^-40 .. 2^40
        s = (torch.rand((rows, 2), generator=g) * 3 + 0.5).to(torch.float8_e4m3fn).to(torch.float32)
        gs = torch.tensor([gsval], dtype=torch.float32)
        got = cta.codec.fp4_quantize_and_pack(d(x, dev), d(s, dev), d(gs, dev), group_size=16)
        assert torch.equal(got.cpu(), O.fp4_compress(x, s, gs, fmt="nvfp4-pack-quantized")["weight_packed"]), gsval
    for sdt in (xdt, F32):  # in-dtype quotient and float32 quotient; powers of two and not
        lo, hi = (-14, 15) if sdt == F16 else (-110, 110)
        e = torch.randint(lo, hi, (rows, 1), generator=g).double()
        for mant in (1.0, 1.5):
            s = (mant * torch.exp2(e)).to(sdt)
            got = cta.codec.fp4_quantize_and_pack(d(x, dev), d(s, dev), None, group_size=32)
            assert torch.equal(got.cpu(), O.fp4_compress(x, s, None, fmt="mxfp4-pack-quantized")["weight_packed"]), (sdt, mant)


@pytest.mark.parametrize("fmt,group", [("nvfp4-pack-quantized", 16), ("mxfp4-pack-quantized", 32)])
def test_fp4_model_compressor_roundtrip(cta, dev, fmt, group):
    """ModelCompressor infers the FP4 format per module, compresses, and the first forward decompresses"""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 512, bias=False), torch.nn.ReLU(), torch.nn.Linear(512, 128, bias=False)).to(dev).to(BF16)
    expect = {}
    for name, m in model.named_modules():
        if not isinstance(m, torch.nn.Linear):
            continue
        m.quantization_scheme = _fp4_scheme(cta, fmt)
        w = m.weight.data.cpu()
        amax = w.float().reshape(w.shape[0], -1, group).abs().amax(-1).clamp(min=1e-4)
        if fmt.startswith("nvfp4"):
            gs = torch.tensor([448.0 * 6.0 / float(amax.max())], dtype=torch.float32)
            s = (gs * amax / 6.0).to(torch.float8_e4m3fn).to(torch.float32)
            m.register_parameter("weight_global_scale", torch.nn.Parameter(gs.to(dev), requires_grad=False))
        else:
            gs, s = None, torch.exp2(torch.floor(torch.log2(amax)) - 2).to(BF16)
        m.register_parameter("weight_scale", torch.nn.Parameter(s.to(dev), requires_grad=False))
        expect[name] = O.fp4_decompress(O.fp4_compress(w, s, gs, fmt=fmt), fmt=fmt)["weight"]
    mc = cta.ModelCompressor()
    mc.compress_model(model)
    for m in model:
        if isinstance(m, torch.nn.Linear):
            assert m.weight_packed.dtype == torch.uint8 and m.quantization_scheme.format.value == fmt
            assert m.weight_scale.dtype == (torch.float8_e4m3fn if fmt.startswith("nvfp4") else torch.uint8)
    y = model(torch.randn(4, 256, device=dev, dtype=BF16))
    assert y.shape == (4, 128)
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.Linear):
            assert eq(m.weight.data.cpu(), expect[name]) and m.weight_scale.dtype == BF16


# ----------------------------------------------------------------------------- FP8 (float-quantized / mxfp8-quantized)
from test_oracle_golden import _f8, _fp8_kw, eq_f8  # noqa: E402

F8 = torch.float8_e4m3fn


def _fp8_args(cta, a, **extra):
    return cta.QuantizationArgs(num_bits=8, type="float", symmetric=True, strategy=a["strategy"], group_size=a.get("group_size"),
                                block_structure=a.get("block_structure"), **extra)


@pytest.mark.parametrize("case", cases("fp8"), ids=lambda c: c["key"])
def test_fp8_quant_golden(golden, cta, dev, case):
    """quantize / dequantize / fake_quantize with FLOAT 8-bit args against the reference's outputs"""
    from compressed_tensors_amd.quantization import dequantize, fake_quantize, quantize

    t = golden.case("fp8", case["key"])
    args = _fp8_args(cta, case["args"])
    zp = _f8(t["zp"]) if case["zp_dtype"] == "float8_e4m3fn" else t["zp"]
    x, s, z = d(t["x"], dev), d(t["scale"], dev), d(zp, dev)
    q = quantize(x, s, z, args, dtype=F8)
    assert q.dtype == F8 and eq_f8(q.cpu(), t["q"])
    assert eq_f8(quantize(x, s, None, args, dtype=F8).cpu(), t["q_nozp"])
    assert eq(quantize(x, s, z, args).cpu(), t["qf"])
    assert eq(fake_quantize(x, s, z, args).cpu(), t["fq"])
    assert eq(dequantize(d(_f8(t["q"]), dev), s, z, args=args).cpu(), t["dq"])
    if case["args"]["strategy"] != "block":
        assert eq(dequantize(d(_f8(t["q"]), dev), s, z).cpu(), t["dq_inferred"])


@pytest.mark.parametrize("case", cases("fp8", "codecs"), ids=lambda c: c["key"])
def test_fp8_codecs_golden(golden, cta, dev, case):
    t = golden.case("fp8", case["key"])
    fmt = case["format"]
    extra = dict(scale_dtype=torch.uint8) if fmt == "mxfp8-quantized" else {}
    args = _fp8_args(cta, case["args"], **extra)
    act = cta.QuantizationArgs(num_bits=8, type="float", strategy="tensor", symmetric=True) if fmt == "float-quantized" else None
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args, input_activations=act)
    comp = cta.BaseCompressor.get_value_from_registry(fmt)
    assert comp.can_compress(torch.nn.Linear, scheme)
    sd = {k[3:]: d(_f8(v) if k.endswith("zero_point") else v, dev) for k, v in t.items() if k.startswith("in.")}
    c = comp.compress(sd, scheme)
    assert sorted(c) == case["compressed_keys"]
    assert {k: str(v.dtype).split(".")[-1] for k, v in c.items()} == case["compressed_dtypes"]
    assert eq_f8(c["weight"].cpu(), t["c.weight"])
    assert torch.equal(c["weight_scale"].cpu().view(torch.uint8) if c["weight_scale"].dtype == torch.uint8 else c["weight_scale"].cpu(), t["c.weight_scale"])
    back = comp.decompress(c, scheme)
    assert sorted(back) == case["decompressed_keys"]
    assert {k: str(v.dtype).split(".")[-1] for k, v in back.items()} == case["decompressed_dtypes"]
    assert eq(back["weight"].cpu(), t["d.weight"]) and eq(back["weight_scale"].cpu(), t["d.weight_scale"])


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_fp8_hardware_conversion_all_inputs(cta, dev, xdt):
    """v_cvt_pk_fp8_f32 / v_cvt_pk_f32_fp8 against the oracle's float8_e4m3fn cast: every 16-bit input, through the flat
    kernel (16-bit scale) and the generic one (float32 scale), with and without a zero point"""
    x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(xdt).reshape(256, 256)
    g = torch.Generator().manual_seed(7)
    for sdt in (xdt, F32):
        for strategy, shape, gs in (("tensor", (1,), None), ("channel", (256, 1), None), ("group", (256, 8), 32)):
            s = (torch.rand(shape, generator=g) * 4 + 0.01).to(sdt)
            if strategy == "tensor":
                s = torch.ones(shape, dtype=sdt)
            for z in (None, torch.zeros(shape, dtype=F8)):
                kw = dict(num_bits=8, strategy=strategy, group_size=gs, qtype="float")
                got = cta.codec.quantize_tensor(d(x, dev), d(s, dev), d(z, dev), dtype=F8, **kw)
                ref = O.quantize(x, s, z, dtype=F8, **kw)
                assert eq_f8(got.cpu(), ref), (sdt, strategy, z is None)
                fq = cta.codec.fake_quantize_tensor(d(x, dev), d(s, dev), d(z, dev), **kw)
                assert eq(fq.cpu(), O.fake_quantize(x, s, z, **kw)), (sdt, strategy, z is None)
    codes = torch.arange(256, dtype=torch.uint8).repeat(64).reshape(64, 256).view(F8)
    for sdt in (BF16, F16, F32):
        s = (torch.rand((64, 8), generator=g) * 4 + 0.01).to(sdt)
        got = cta.codec.dequantize_tensor(d(codes, dev), d(s, dev), None)
        assert eq(got.cpu(), O.dequantize(codes, s, None))


def test_fp8_of_float32_weights_flat_kernels(cta, dev):
    """float32 weights -> float8_e4m3fn and back on the flat float32 kernels (f32_quant_units_kernel<.., FP8>, f32_quads_kernel<F32_DQ, .., FP8>): quotients that
    sit exactly on a float8 rounding boundary (x = midpoint * s for scales with a short significand) and one / two ulps either side, the float8 subnormal
    range, +-448 and beyond, inf / NaN / signed zeros / float32 subnormals, scales from 2^-110 to 2^110, float32 and bf16 scales, every strategy the flat
    kernels take (tensor, channel, group, block), with no zero point, an all-zero one and a non-zero one; the dequantize side over every float8 code"""
    g = torch.Generator().manual_seed(91)
    rows, cols = 128, 1024
    codes = torch.arange(256, dtype=torch.uint8).view(F8).float()
    vals = codes[torch.isfinite(codes)].unique()
    mids = (vals[:-1] + vals[1:]) / 2  # every rounding boundary of float8_e4m3fn (exact in float32)
    pool = torch.cat([mids, vals, torch.tensor([448.0, -448.0, 464.0, -464.0, 447.99997, 2.0 ** -10, -(2.0 ** -10), 3 * 2.0 ** -10])])
    scales = torch.tensor([0.125, 0.0390625, 3.0, 1.0 / 3.0, 0.1, 7.3e-5, 2.0 ** -90, 2.0 ** 90, 2.0 ** -110, 2.0 ** 110, 1e-3, 0.75, 5.0, 1.7, 2.0 ** -20, 9.5e4], dtype=F32)
    s_row = scales[torch.arange(rows) % scales.numel()].reshape(rows, 1).clone()
    x = (pool[torch.randint(0, pool.numel(), (rows, cols), generator=g)] * s_row).contiguous()
    xb = x.view(torch.int32)
    xb += torch.randint(-2, 3, (rows, cols), generator=g, dtype=torch.int32)
    x[:, -32:] = torch.randn(rows, 32, generator=g) * s_row * 100
    sp = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 3e38, -3e38, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38], dtype=F32)
    x[:, : sp.numel()] = sp
    for sdt in (F32, BF16):
        for strategy, shape, gs, block in (("channel", (rows, 1), None, None), ("tensor", (1,), None, None), ("group", (rows, cols // 128), 128, None),
                                           ("block", (rows // 64, cols // 128), None, [64, 128])):
            if strategy == "channel":
                s = s_row.to(sdt)
            elif strategy == "tensor":
                s = torch.tensor([0.37], dtype=sdt)
            elif strategy == "group":
                s = s_row.expand(rows, cols // 128).contiguous().to(sdt)
            else:
                s = (torch.rand(shape, generator=g) * 3 + 0.01).to(sdt)
            kw = dict(num_bits=8, strategy=strategy, group_size=gs, block_structure=block, qtype="float")
            for z in (None, torch.zeros(s.shape, dtype=F8), (torch.randn(s.shape, generator=g) * 4).to(F8)):
                got = cta.codec.quantize_tensor(d(x, dev), d(s, dev), d(z, dev), dtype=F8, **kw)
                ref = O.quantize(x, s, z, dtype=F8, **kw)
                assert eq_f8(got.cpu(), ref), (sdt, strategy, None if z is None else float(z.float().abs().max()))
    allc = torch.arange(256, dtype=torch.uint8).repeat(rows * cols // 256).reshape(rows, cols).view(F8)
    for strategy, s, gs, block in (("channel", s_row, None, None), ("group", (torch.rand((rows, cols // 128), generator=g) * 4 - 2), 128, None),
                                   ("tensor", torch.tensor([-0.37]), None, None), ("block", torch.rand((rows // 64, cols // 128), generator=g) + 0.5, None, [64, 128])):
        for z in (None, (torch.randn(s.shape, generator=g) * 4).to(F8)):
            dkw = dict(strategy=strategy, group_size=gs, block_structure=block)
            got = cta.codec.dequantize_tensor(d(allc, dev), d(s, dev), d(z, dev), **dkw)
            ref = O.dequantize(allc, s, z, **dkw)
            assert got.dtype == F32 and eq(got.cpu(), ref), (strategy, z is None)


def test_fp8_quantize_of_fp16_weights_every_input_times_scales(cta, dev):
    """the packed-fp16 form of the FP8 quantize (f8_quant_words_f16: reciprocal + Newton step on pairs) and its per-unit precondition (finite, zero or a
    quotient of at least 2^-13): EVERY fp16 input against 96 channel scales — the ends of the fast range 2^-14 / 2^15 and just outside it, powers of two,
    scales that put subnormal inputs on either side of the 2^-13 line, random ones — with the compressors' all-zero zero point and without one, bit for bit (the sign of a
    zero result included) against the oracle; sorted and shuffled inputs (a unit of 8 mixes tiny / non-finite elements with ordinary ones)"""
    g = torch.Generator().manual_seed(3)
    x0 = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(F16)
    fixed = [2.0 ** -14, 2.0 ** -14 * 1.001, 2.0 ** -15, 2.0 ** 15, 2.0 ** 15 * 0.999, 65504.0, 2.0 ** -11, 2.0 ** -10, 2.0 ** -9, 1.0, -1.0, 448.0, 1 / 448.0, 3.0, -0.37]
    scales = torch.cat([torch.tensor(fixed), 2.0 ** torch.linspace(-14, 15, 49), torch.rand(32, generator=g) * 2.0 ** torch.randint(-14, 4, (32,), generator=g).float()])
    s = scales.to(F16).reshape(-1, 1)
    s[s == 0] = 1.0
    kw = dict(num_bits=8, strategy="channel", qtype="float")
    for x1 in (x0, x0[torch.randperm(65536, generator=g)]):
        x = x1.reshape(1, -1).repeat(s.shape[0], 1).contiguous()
        for z in (torch.zeros(s.shape, dtype=F8), None):  # without a zero point a -0 weight stays -0
            got = cta.codec.quantize_tensor(d(x, dev), d(s, dev), d(z, dev), dtype=F8, **kw)
            assert eq_f8(got.cpu(), O.quantize(x, s, z, dtype=F8, **kw)), z is None
        # the one-pass channel-wise compress shares the word builder (its own scale: amax / 448)
        q, sc, _ = cta.codec.rtn_quantize_channel8(d(x[:8, :16384].contiguous(), dev), qtype="float", symmetric=True)
        s_ref = O.calculate_qparams_float(x[:8, :16384], kind="fp8")
        assert eq(sc.cpu(), s_ref) and eq_f8(q.cpu(), O.quantize(x[:8, :16384], s_ref, torch.zeros_like(s_ref, dtype=F8), dtype=F8, **kw))


def test_fp8_full_size_roundtrip(cta, dev):
    """8192 x 8192 bf16, channel scales: dequantize(quantize(x)) is idempotent under a second round trip and every
    code is a finite float8 value"""
    N = 8192
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, N), generator=g, device=dev, dtype=BF16)
    s = (x.abs().amax(dim=1, keepdim=True).float() / 448.0).to(BF16)
    kw = dict(num_bits=8, strategy="channel", qtype="float")
    q = cta.codec.quantize_tensor(x, s, None, dtype=F8, **kw)
    y = cta.codec.dequantize_tensor(q, s, None)
    assert not bool(torch.isnan(y).any()) and y.dtype == BF16
    q2 = cta.codec.quantize_tensor(y, s, None, dtype=F8, **kw)
    y2 = cta.codec.dequantize_tensor(q2, s, None)
    assert torch.equal(y2, y)
    rel = ((y.float() - x.float()).abs() / s.float()).max()
    # half a float8 step at the top binade (32 / 2) + the clamp excess of a bf16-rounded scale (448 * 2^-8) + bf16 rounding of y
    assert float(rel) <= 20.0


@pytest.mark.parametrize("sdt", [BF16, F16, F32])
def test_dequantize_keeps_the_sign_of_a_zero_product(cta, dev, sdt):
    """0 * (negative scale) = -0.0 and (-0.0 code) * scale = -0.0, in every scale dtype (fp16 products are not fused
    into a +0 accumulate)"""
    q = torch.zeros((4, 64), dtype=torch.int8)
    s = torch.full((4, 1), -0.5, dtype=sdt)
    got = cta.codec.dequantize_tensor(d(q, dev), d(s, dev), None)
    assert eq(got.cpu(), O.dequantize(q, s, None)) and bool(torch.signbit(got).all())
    q8 = torch.full((4, 64), 0x80, dtype=torch.uint8).view(F8)
    s = torch.full((4, 1), 0.5, dtype=sdt)
    got = cta.codec.dequantize_tensor(d(q8, dev), d(s, dev), None)
    assert eq(got.cpu(), O.dequantize(q8, s, None)) and bool(torch.signbit(got).all())


# ----------------------------------------------------------------------------- one-pass round-to-nearest compress (N1 fused)
@pytest.mark.parametrize("xdt", [BF16, F16])
@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("shape,gs", [((8, 512), 128), ((5, 256), 32), ((3, 2048), 2048), ((64, 4096), 128), ((7, 1024), None), ((16, 64), 64)])
def test_rtn_one_pass_equals_observer_plus_compress(cta, dev, xdt, symmetric, shape, gs):
    """ct_rtn_quant_pack_w4 against the oracle's calculate_qparams + pack-quantized compress, and against the two-kernel
    composition on the device; weights carry special values, an all-zero group and a strictly positive group"""
    g = torch.Generator().manual_seed(shape[0] * 7 + shape[1])
    x = (torch.randn(shape, generator=g) * 0.05).to(xdt)
    group = gs or shape[1]
    sp = special_values(xdt)
    sp = sp[torch.isfinite(sp.float())]
    x.view(-1)[: min(sp.numel(), x.numel())] = sp[: x.numel()]
    if shape[0] > 2:
        x[1, :group] = 0
        x[2, :group] = x[2, :group].abs() + 0.01
    packed, scale, zp = cta.codec.rtn_quantize_and_pack(d(x, dev), group_size=gs, symmetric=symmetric)
    s_ref, z_ref = O.calculate_qparams_minmax(x, num_bits=4, group_size=gs, symmetric=symmetric)
    assert eq(scale.cpu(), s_ref) and torch.equal(zp.cpu(), z_ref)
    strategy = "group" if gs else "channel"
    q = O.quantize(x, s_ref, z_ref, num_bits=4, strategy=strategy, group_size=gs, dtype=torch.int8)
    assert torch.equal(packed.cpu(), O.pack_to_int32(q, 4).contiguous())
    s2, z2 = cta.codec.minmax_qparams(d(x, dev), num_bits=4, group_size=gs, symmetric=symmetric)
    p2 = cta.codec.quantize_and_pack(d(x, dev), s2, z2, num_bits=4, strategy=strategy, group_size=gs)
    assert torch.equal(packed, p2) and eq(scale.cpu(), s2.cpu()) and torch.equal(zp, z2)


def test_rtn_nan_group_and_class_api(cta, dev):
    x = torch.randn((4, 256), dtype=BF16)
    x[0, 3] = float("nan")
    packed, scale, zp = cta.codec.rtn_quantize_and_pack(d(x, dev), group_size=128, symmetric=False)
    s_ref, z_ref = O.calculate_qparams_minmax(x, num_bits=4, group_size=128, symmetric=False)
    assert eq(scale.cpu(), s_ref) and torch.equal(zp.cpu(), z_ref)
    q = O.quantize(x, s_ref, z_ref, num_bits=4, strategy="group", group_size=128, dtype=torch.int8)
    assert torch.equal(packed.cpu(), O.pack_to_int32(q, 4).contiguous())
    for sym, gs, strategy in ((True, 128, "group"), (False, 128, "group"), (True, None, "channel"), (True, 48, "group")):
        shape = (8, 96 if gs == 48 else 512)
        w = torch.randn(shape, dtype=BF16, device=dev)
        args = cta.QuantizationArgs(num_bits=4, group_size=gs, symmetric=sym, strategy=strategy)
        scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
        got = cta.PackedQuantizationCompressor.compress_rtn(w, scheme)
        s2, z2 = cta.codec.minmax_qparams(w, num_bits=4, group_size=gs, symmetric=sym)
        ref = cta.PackedQuantizationCompressor.compress({"weight": w, "weight_scale": s2, "weight_zero_point": z2}, scheme)
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert torch.equal(got[k].cpu(), ref[k].cpu()), (sym, gs, k)


def test_rtn_full_size(cta, dev):
    N = 8192
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((N, N), generator=g, device=dev, dtype=BF16)
    packed, scale, zp = cta.codec.rtn_quantize_and_pack(x, group_size=128, symmetric=True)
    s2, z2 = cta.codec.minmax_qparams(x, num_bits=4, group_size=128, symmetric=True)
    p2 = cta.codec.quantize_and_pack(x, s2, z2, num_bits=4, strategy="group", group_size=128)
    assert torch.equal(packed, p2) and torch.equal(scale.view(torch.int16), s2.view(torch.int16)) and torch.equal(zp, z2)


# ----------------------------------------------------------------------------- FLOAT 4-bit quantize / dequantize / fake_quantize
@pytest.mark.parametrize("case", cases("fp4q"), ids=lambda c: c["key"])
def test_fp4_quant_golden(golden, cta, dev, case):
    """quantize / fake_quantize / dequantize with FLOAT 4-bit args (NVFP4's tensor_group + global scale, MXFP4's group 32,
    plain group / channel / tensor) against the reference's outputs"""
    from compressed_tensors_amd.quantization import dequantize, fake_quantize, quantize

    t = golden.case("fp4q", case["key"])
    a = case["args"]
    args = cta.QuantizationArgs(num_bits=4, type="float", symmetric=True, strategy=a["strategy"], group_size=a.get("group_size"))
    zp = _f8(t["zp"]) if case["zp_dtype"] == "float8_e4m3fn" else t["zp"]
    x, s, z, gs = d(t["x"], dev), d(t["scale"], dev), d(zp, dev), d(t.get("gs"), dev)
    assert eq(quantize(x, s, z, args, global_scale=gs).cpu(), t["qf"])
    assert eq(quantize(x, s, None, args, global_scale=gs).cpu(), t["qf_nozp"])
    assert eq(fake_quantize(x, s, z, args, global_scale=gs).cpu(), t["fq"])
    assert eq(dequantize(d(t["qf"], dev), s, z, args=args, global_scale=gs).cpu(), t["dq"])


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_fp4_quantize_all_inputs(cta, dev, xdt):
    """every 16-bit input through the generic FLOAT 4-bit path: in-dtype scales (fq16 fast path and the unit kernel) and
    float32 scales under a global scale"""
    x = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(xdt).reshape(256, 256)
    g = torch.Generator().manual_seed(9)
    for sdt, gsval in ((xdt, None), (F32, None), (F32, 37.5)):
        s = (torch.rand((256, 16), generator=g) * 2 + 0.05).to(sdt)
        gs = None if gsval is None else torch.tensor([gsval], dtype=F32)
        for z in (None, torch.zeros((256, 16), dtype=F8)):
            kw = dict(num_bits=4, strategy="group", group_size=16, qtype="float", global_scale=gs)
            got = cta.codec.quantize_tensor(d(x, dev), d(s, dev), d(z, dev), **{**kw, "global_scale": d(gs, dev)})
            assert eq(got.cpu(), O.quantize(x, s, z, **kw)), (sdt, gsval, z is None)
            fq = cta.codec.fake_quantize_tensor(d(x, dev), d(s, dev), d(z, dev), **{**kw, "global_scale": d(gs, dev)})
            assert eq(fq.cpu(), O.fake_quantize(x, s, z, **kw)), (sdt, gsval, z is None)


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """the N > 1 path of bench.py (launch contract, barrier, max-over-ranks timing, rank-sharded checkpoint leg) with
    both ranks on cuda:0 and the timing collectives over gloo (CT_BENCH_SHARE_GPU=1)"""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CT_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak" and out["parity_gate"] is True
    assert out["value"] > 0 and out["roofline"]["frac"] > 0 and "cpu_baseline" not in out
    leg = out["tinyllama_checkpoint"]
    assert leg["round_trip_equals_fake_quantize"] is True and 0 < leg["modules_this_rank"] < 154
    assert out["oracle_slice_check"] is True
    rs = out["row_sharded"]  # ONE tensor split by row blocks over the ranks; each shard is a slice of the single-rank result
    assert rs["ranks"] == 2 and rs["rows_this_rank"] == [0, 4096]
    assert rs["w4a16"]["shard_equals_slice_of_single_rank_result"] is True and rs["w4a16"]["GBps_all_ranks"] > 0
    assert rs["sparse_bitmask"]["shard_equals_slice_of_single_rank_result"] is True


@pytest.mark.parametrize("world", [4, 8])
def test_bench_many_ranks_on_one_gpu(world):
    """VERDICT r05 next #2c: `CT_BENCH_SHARE_GPU=1 python3 bench.py --gpus 8 --steps 6 --warmup 2` (and --gpus 4) — every line of the
    N = 8 code path executes before the driver's first real 8-GPU run: the self-launch, eight ranks in the process group, the LPT split of
    the 154 TinyLlama modules covering each exactly once, the row-sharded legs on the HBM-cold C-ABI protocol (every shard equal to the
    slice of the single-rank result), the 4096^2 leg at this world size, and a final line inside the driver's window"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CT_BENCH_SHARE_GPU="1", CT_BENCH_WARM_SCALE="0.2")  # (the ranks take turns on ONE GPU: shorter device warm-ups, same code path)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) <= 6000, (len(lines), [len(x) for x in lines])
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 6 and out["scaling"] == "weak" and out["parity_gate"] is True and out["oracle_slice_check"] is True
    cfg = out["config"]
    assert cfg["ranks_seen"] == world and len(cfg["per_rank_GBps"]) == world and all(v > 0 for v in cfg["per_rank_GBps"])
    t = out["tinyllama_checkpoint"]
    assert t["every_module_on_exactly_one_rank"] is True and sum(t["modules_per_rank"]) == 154 and len(t["modules_per_rank"]) == world
    assert all(v > 0 for v in t["modules_per_rank"]) and t["round_trip_equals_fake_quantize"] is True and t["rotating_copies"] == world
    rs = out["row_sharded"]
    assert rs["ranks"] == world and rs["rows_this_rank"] == [0, 8192 // world]
    assert rs["w4a16"]["shard_equals_slice_of_single_rank_result"] is True and rs["w4a16"]["sets"] == 16 * world and rs["w4a16"]["GBps_all_ranks"] > 0
    assert rs["w4a16"]["us_per_tensor_hip_graph"] > 0 and out["w4a16_4096"]["us_per_step_hip_graph"] > 0  # the same steps replayed from one captured HIP graph
    assert rs["sparse_bitmask"]["shard_equals_slice_of_single_rank_result"] is True and rs["sparse_bitmask"]["sets"] == 8 * world
    k4 = out["w4a16_4096"]
    assert k4["ranks"] == world and k4["round_trip_equals_fake_quantize"] is True and k4["GBps_all_ranks"] > 0


def test_bench_self_launches_two_ranks_without_a_launcher():
    """exactly `CT_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 6 --warmup 2` — no torchrun in the
    command — must start its two ranks itself, exit 0 and print ONE line with n_gpus 2, the process group's size and what every
    rank did on its own"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CT_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["parity_gate"] is True and out["value"] > 0
    cfg = out["config"]
    assert cfg["ranks_seen"] == 2 and len(cfg["per_rank_GBps"]) == 2 and all(v > 0 for v in cfg["per_rank_GBps"])
    assert cfg["value_one_stream"] > 0


@pytest.mark.parametrize("n,dtype,sym", [(8192, BF16, True), (2048, BF16, False), (2048, F16, True), (2048, F16, False)])
def test_reference_op_sequence_on_the_gpu(cta, dev, n, dtype, sym):
    """the reference's own eager op sequence (oracle/eager_ref.py, pinned bit-for-bit against the reference in the build container)
    executed by PyTorch-ROCm's kernels on THIS GPU — what upstream would compute here — against the fused HIP kernels: packed
    words, packed zero points and decompressed weights bit for bit; also BASELINE config 1 (int8 per tensor)"""
    import eager_ref as E

    torch.manual_seed(n)
    w = torch.randn(n, n, dtype=dtype, device=dev)
    s, z = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=sym)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=sym))
    sd = {"weight": w, "weight_scale": s, "weight_zero_point": z}
    ref_c = E.pack_quantized_compress(sd, num_bits=4, strategy="group", group_size=128, symmetric=sym)
    got_c = cta.PackedQuantizationCompressor.compress(sd, scheme)
    assert sorted(ref_c) == sorted(got_c)
    for k in ref_c:
        assert torch.equal(ref_c[k].contiguous().cpu() if k == "weight_shape" else ref_c[k].contiguous(), got_c[k].cpu() if k == "weight_shape" else got_c[k]), k
    ref_d = E.pack_quantized_decompress(ref_c, num_bits=4, strategy="group", symmetric=sym)
    got_d = cta.PackedQuantizationCompressor.decompress(got_c, scheme)
    assert eq(got_d["weight"].cpu(), ref_d["weight"].cpu())
    if n <= 4096:
        s1 = (w.abs().max().float() / 127.0).to(dtype).reshape(1)
        sd8 = {"weight": w, "weight_scale": s1, "weight_zero_point": torch.zeros(1, dtype=torch.int8, device=dev)}
        a8 = cta.QuantizationArgs(num_bits=8, strategy="tensor", symmetric=True)
        sch8 = cta.QuantizationScheme(targets=["Linear"], weights=a8, input_activations=a8)
        r8, g8 = E.int_quantized_compress(sd8), cta.IntQuantizationCompressor.compress(sd8, sch8)
        assert torch.equal(r8["weight"], g8["weight"])
        assert eq(cta.IntQuantizationCompressor.decompress(g8, sch8)["weight"].cpu(), E.int_quantized_decompress(r8)["weight"].cpu())


def test_rccl_process_group_on_one_gpu(tmp_path):
    """what a one-GPU lease allows of the RCCL path: a world-size-1 `nccl` (= RCCL) process group under torch.distributed.run —
    init_dist's backend choice and device binding, the barrier + MAX all-reduce on device tensors that bench.py uses for timing,
    and ModelCompressor running with `is_distributed()` true (LPT shard = everything, recouple a no-op)"""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import compressed_tensors_amd as cta
from compressed_tensors_amd.distributed import init_dist, is_distributed, rank_and_world
init_dist()
assert is_distributed() and dist.get_backend() == "nccl" and rank_and_world() == (0, 1)
dev = torch.device("cuda", torch.cuda.current_device())
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.barrier(); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 3.5
buf = torch.arange(1024, dtype=torch.uint8, device=dev); dist.broadcast(buf, src=0)
net = torch.nn.Sequential(*[torch.nn.Linear(256, 64, bias=False) for _ in range(3)]).to(dev).to(torch.bfloat16)
args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=False)
ref = []
for m in net:
    m.quantization_scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    s, z = cta.quantization.calculate_qparams_from_weight(m.weight.data, args)
    m.register_parameter("weight_scale", torch.nn.Parameter(s, requires_grad=False))
    m.register_parameter("weight_zero_point", torch.nn.Parameter(z, requires_grad=False))
    ref.append(cta.codec.fake_quantize_tensor(m.weight.data, s, z, num_bits=4, strategy="group", group_size=128))
mc = cta.ModelCompressor()
mine = mc.compress_model(net)
assert len(mine) == 3 and all(hasattr(m, "weight_packed") for m in net) and hasattr(net, "ct_decompress_hook")
mc.decompress_model(net, recouple=True)
assert all(torch.equal(m.weight.data, r) for m, r in zip(net, ref))
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
""" % root
    script = tmp_path / "rccl_one.py"
    script.write_text(code)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


_RECOUPLE_DEVICE_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import compressed_tensors_amd as cta
from compressed_tensors_amd.distributed import is_distributed, rank_and_world
backend = os.environ["CT_TEST_BACKEND"]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)   # BOTH ranks on the one GPU of this box
if backend == "nccl":
    dist.init_process_group("nccl", init_method="env://", device_id=dev)
else:
    dist.init_process_group("gloo", init_method="env://")
rank, world = rank_and_world()
assert is_distributed() and world == 2
probe = torch.full((1 << 20,), rank + 1, dtype=torch.uint8, device=dev)
dist.broadcast(probe, src=1)   # a device buffer really moves between the two processes
assert int(probe[0]) == 2 and int(probe[-1]) == 2

def model(sym):
    torch.manual_seed(0)   # the same replica on both ranks
    net = torch.nn.Sequential(*[torch.nn.Linear(256 * (1 + i %% 3), 64 * (1 + i), bias=False) for i in range(7)]).to(dev).to(torch.bfloat16)
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=sym)
    fq, packed = [], []
    for m in net:
        m.quantization_scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
        s, z = cta.quantization.calculate_qparams_from_weight(m.weight.data, args)
        m.register_parameter("weight_scale", torch.nn.Parameter(s, requires_grad=False))
        m.register_parameter("weight_zero_point", torch.nn.Parameter(z, requires_grad=False))
        fq.append(cta.codec.fake_quantize_tensor(m.weight.data, s, z, num_bits=4, strategy="group", group_size=128))
        packed.append(cta.codec.quantize_and_pack(m.weight.data, s, z, num_bits=4, strategy="group", group_size=128))
    return net, fq, packed

moved = 0
real_broadcast = dist.broadcast
def counting_broadcast(t, src, *a, **k):
    global moved
    if t.is_cuda:
        moved += t.numel() * t.element_size()
    return real_broadcast(t, src, *a, **k)
import compressed_tensors_amd.distributed.module_parallel as mp
mp.dist.broadcast = counting_broadcast

for sym in (True, False):
    net, fq, packed = model(sym)
    mc = cta.ModelCompressor()
    mine = mc.compress_model(net)   # default: every rank ends with the whole model compressed (reference module_parallel.py:59-90)
    assert 0 < len(mine) < 7, len(mine)
    for m, p in zip(net, packed):
        assert hasattr(m, "weight_packed") and not hasattr(m, "weight") and m.weight_packed.is_cuda and torch.equal(m.weight_packed.data, p)
        assert isinstance(m._parameters["weight_packed"], torch.nn.Parameter) and not m.weight_packed.requires_grad
        if not sym:
            assert m.weight_zero_point.dtype == torch.int32   # travels packed, as upstream stores it
    got = torch.tensor([float(len(mine))], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(got)
    assert int(got.item()) == 7   # the two shares partition the model
    mc.decompress_model(net, recouple=True)   # the distributed decompress upstream leaves as a TODO (model_compressor.py:196-198)
    for m, r in zip(net, fq):
        assert m.weight.is_cuda and torch.equal(m.weight.data, r)
assert moved > 1 << 20, moved   # the flat buffers were DEVICE bytes
dist.barrier()
dist.destroy_process_group()
print("RECOUPLE_OK", backend, rank, moved)
"""


def test_world_size_2_recouple_moves_device_bytes(tmp_path):
    """N3 on hardware (VERDICT r05 next #2d): two ranks, both on cuda:0, the REAL codecs — ModelCompressor.compress_model with the
    default recouple (each rank compresses its LPT share with the batched HIP kernels, then one flat DEVICE buffer per owner is
    broadcast and the receivers' parameters are views into it) and the distributed decompress — over RCCL when RCCL accepts two ranks
    on one device, else over gloo with device tensors (the collective still takes and delivers device memory).  Every packed word equal
    to the single-process result, every decompressed weight equal to fake_quantize, on both ranks."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "recouple_device.py"
    script.write_text(_RECOUPLE_DEVICE_SCRIPT % {"root": root})
    tried = {}
    for backend in ("nccl", "gloo"):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", CT_TEST_BACKEND=backend)
        try:
            r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                                str(port), str(script)], capture_output=True, text=True, timeout=420, env=env)
        except subprocess.TimeoutExpired as e:
            tried[backend] = f"timeout: {str(e.stderr)[-600:]}"
            continue
        ok = r.returncode == 0 and r.stdout.count("RECOUPLE_OK") == 2
        tried[backend] = "ok" if ok else (r.stdout[-800:] + r.stderr[-2500:])
        if ok:
            break
        if backend == "nccl":  # RCCL refuses two ranks on one device ("Duplicate GPU detected"): that, and only that, sends the test to gloo
            text = r.stdout + r.stderr
            assert any(sig in text for sig in ("Duplicate GPU", "duplicate", "invalid usage", "ncclInvalidUsage", "NCCL error", "ncclUnhandled", "ProcessGroupNCCL")), text[-3000:]
    assert "ok" in tried.values(), tried
    print("recouple backends:", {k: (v if v == "ok" else v[-200:]) for k, v in tried.items()})


@pytest.mark.parametrize("wdt", [BF16, F16])
@pytest.mark.parametrize("bits", [4, 8])
def test_marlin24_front_end_special_values(cta, dev, wdt, bits):
    """the packed-fp16 back end of the fused front end against quantize (fp16) + cutlass 2:4 compress: kept values that are
    inf / NaN / huge / fp16-subnormal / on rounding ties, scales at both ends of the fast range and outside it, a non-zero
    zero point, every fp16 value once"""
    g = torch.Generator().manual_seed(17 + bits)
    rows, cols = 128, 1024
    # every 16-bit pattern of the weight dtype in the kept slots of a 2:4 layout (two values, two zeros)
    allv = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(wdt)
    w = torch.zeros((rows, cols), dtype=wdt)
    w.view(-1, 4)[:, 0] = allv[: rows * cols // 4]
    w.view(-1, 4)[:, 2] = allv[rows * cols // 4: rows * cols // 2] if rows * cols // 2 <= 65536 else 0
    half = 2.0 ** (bits - 1)
    ties = (torch.arange(-2 * half - 2, 2 * half + 2, dtype=torch.float32) + 0.5)
    for case, smaker in (("mid", lambda: (torch.rand((rows, cols // 128), generator=g) * 0.2 + 0.01)),
                         ("tiny", lambda: torch.full((rows, cols // 128), 2.0 ** -14)), ("big", lambda: torch.full((rows, cols // 128), 2.0 ** 15)),
                         ("subnormal", lambda: torch.full((rows, cols // 128), 2.0 ** -20)), ("ones", lambda: torch.full((rows, cols // 128), 1.0))):
        s = smaker().to(F16)
        ww = w.clone()
        if case == "ones":
            ww.view(-1, 4)[: ties.numel(), 1] = ties.to(wdt)  # exact .5 ties under a unit scale
            ww.view(-1, 4)[: ties.numel(), 0] = 0
            ww.view(-1, 4)[: ties.numel(), 2] = 0
        for z in (torch.zeros_like(s, dtype=torch.int8), torch.full(s.shape, 3, dtype=torch.int8)):
            comp, meta, bad = cta.codec.marlin24_quant_compress(d(ww, dev), d(s, dev), d(z, dev), num_bits=bits, group_size=128)
            q = cta.codec.quantize_tensor(d(ww, dev).to(F16), d(s, dev), d(z, dev), num_bits=bits, strategy="group", group_size=128)
            ok24 = bool(((q != 0).view(-1, 4).sum(-1) <= 2).all())
            assert bool(bad.item()) == (not ok24), (case, int(z[0, 0]))
            if ok24:
                comp_ref, meta_ref = cta.codec.cutlass24_from_dense(q)
                # a NaN is kept as a non-zero by the 2:4 selection and becomes code 0 in the integer cast (as the exact form)
                assert torch.equal(comp.cpu().float(), torch.nan_to_num(comp_ref.cpu().float(), nan=0.0)) and torch.equal(meta.cpu(), meta_ref.cpu()), (case, int(z[0, 0]))


@pytest.mark.parametrize("sdt", [BF16, F16])
@pytest.mark.parametrize("wdt", [BF16, F16])
def test_marlin24_lean_w4_special_values(cta, dev, wdt, sdt):
    """the one-launch int4 kernel (lean front end: raw bf16 -> fp32, sum-of-squares range test, packed fp32 quotient, one multiply
    for bf16 / bf16, dot4 + v_perm table selection) against the unfused chain it replaces (ct_marlin24_quant_compress, pinned
    above, + ct_marlin24_pack_weights): every 16-bit weight pattern in the kept slots (inf, NaN, huge, sub-fp16-normal, ties),
    scales inside, at both ends of and outside the lean range [2^-12, 2^15], a zero / non-zero / absent zero point"""
    g = torch.Generator().manual_seed(23)
    rows, cols = 128, 1024
    allv = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(wdt)
    w = torch.zeros((rows, cols), dtype=wdt)
    w.view(-1, 4)[:, 0] = allv[: rows * cols // 4]
    w.view(-1, 4)[:, 3] = allv[rows * cols // 4: rows * cols // 2]
    ties = (torch.arange(-18, 18, dtype=torch.float32) + 0.5)
    G = cols // 128
    cases = {"mid": lambda: torch.rand((rows, G), generator=g) * 0.2 + 0.01, "ones": lambda: torch.full((rows, G), 1.0),
             "2^-12": lambda: torch.full((rows, G), 2.0 ** -12), "2^-13": lambda: torch.full((rows, G), 2.0 ** -13),
             "2^-14": lambda: torch.full((rows, G), 2.0 ** -14), "2^15": lambda: torch.full((rows, G), 2.0 ** 15),
             "1.5*2^15": lambda: torch.full((rows, G), 1.5 * 2.0 ** 15), "negative": lambda: -(torch.rand((rows, G), generator=g) + 0.5),
             "mixed": lambda: torch.where(torch.rand((rows, G), generator=g) < 0.5, torch.full((rows, G), 2.0 ** -13), torch.full((rows, G), 0.37))}
    if sdt == F16:
        cases["subnormal"] = lambda: torch.full((rows, G), 2.0 ** -20)
    else:
        cases["huge"] = lambda: torch.full((rows, G), 2.0 ** 40)   # bf16 scale whose fp16 image is inf
        cases["vanishing"] = lambda: torch.full((rows, G), 2.0 ** -40)  # ... is zero
    for case, smaker in cases.items():
        s = smaker().to(sdt)
        ww = w.clone()
        if case == "ones":
            ww.view(-1, 4)[: ties.numel(), 1] = ties.to(wdt)
            ww.view(-1, 4)[: ties.numel(), 0] = 0
            ww.view(-1, 4)[: ties.numel(), 3] = 0
        zmix = torch.zeros(s.shape, dtype=torch.int8)
        zmix[::3, 1::2] = 2
        for z in (torch.zeros(s.shape, dtype=torch.int8), None, zmix):
            packed, meta, bad = cta.codec.marlin24_compress_w4(d(ww, dev), d(s, dev), d(z, dev), group_size=128)
            comp, meta_ref, bad_ref = cta.codec.marlin24_quant_compress(d(ww, dev), d(s, dev), d(z, dev), num_bits=4, group_size=128)
            packed_ref = cta.codec.marlin24_pack_weights(comp, 4, transposed=True, add_offset=True)
            tag = (case, None if z is None else int(z.abs().sum() > 0))
            assert bool(bad.item()) == bool(bad_ref.item()), tag
            if not bool(bad_ref.item()):
                assert torch.equal(meta.cpu(), meta_ref.cpu()), tag
                assert torch.equal(packed.cpu(), packed_ref.cpu()), tag


def test_marlin24_lean_quotients_are_exact(cta):
    """ct_selftest_m24_div: fp16 / fp16 by reciprocal + Newton step and bf16 / bf16 by ONE multiply give the fp16 rounding of
    the IEEE quotient for every scale in the lean range x all 65536 weights"""
    assert cta.codec.selftest_m24_div(0) == 0
    assert cta.codec.selftest_m24_div(1) == 0
    assert cta.codec.selftest_m24_div(2) == 0  # round 5: v_rcp_f32 + two Newton steps == the IEEE `1.0f / s` for every fp16 scale, bit for bit


def test_marlin24_deferred_structure_check(cta, dev):
    """inside `deferred_structure_check()` compress only queues work; ONE ValueError at the exit reports any non-2:4 weight of the
    batch, the flag ring is clean afterwards, and outside the context the error is raised by the call itself (as upstream)"""
    M = cta.Marlin24Compressor
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, strategy="group", group_size=128, symmetric=True))

    def sd(sparse, seed):
        g = torch.Generator().manual_seed(seed)
        w = torch.randn((64, 512), generator=g).to(BF16)
        if sparse:
            w = w * O.sparse24_mask(w).to(w.dtype)
        scale, zp = O.calculate_qparams_minmax(w.to(F16), num_bits=4, group_size=128, symmetric=True)
        return {"weight": w.to(dev), "weight_scale": scale.to(dev), "weight_zero_point": zp.to(dev)}

    good = [sd(True, i) for i in range(3)]
    with M.deferred_structure_check():
        outs = [M.compress(x, scheme) for x in good]
    with pytest.raises(ValueError, match="2:4 sparsity structure"):
        with M.deferred_structure_check():
            M.compress(good[0], scheme)
            M.compress(sd(False, 9), scheme)  # no exception here ...
            late = M.compress(good[1], scheme)  # ... and later calls still run
    assert torch.equal(late["weight_packed"], outs[1]["weight_packed"])
    with M.deferred_structure_check():  # the ring was released: a clean batch passes again
        again = M.compress(good[2], scheme)
    assert torch.equal(again["weight_packed"], outs[2]["weight_packed"]) and torch.equal(again["meta"], outs[2]["meta"])
    with pytest.raises(ValueError, match="2:4 sparsity structure"):
        M.compress(sd(False, 10), scheme)
    M.compress(good[0], scheme)


@pytest.mark.parametrize("shape", [(64, 256), (192, 768), (4096, 1024), (2048, 8192)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_marlin24_verdict_word_through_the_c_abi(cta, dev, shape):
    """`ct_marlin24_compress_w4_verdict` (round 5): the launch's last-reporting workgroup stores 1 (2:4 holds) or 3 (violated) into the
    caller's pinned word — one workgroup, a grid below / not a multiple of / far above the 64 leaves of the ticket tree; the outputs are
    those of `ct_marlin24_compress_w4_full`; thirty launches in a row (the trees return to zero) alternate the two verdicts"""
    from compressed_tensors_amd import _lib

    lib = _lib.load()
    m, k = shape
    g = torch.Generator(device=dev).manual_seed(m + k)
    w = torch.randn(m, k, dtype=BF16, device=dev, generator=g)
    w = w * cta.codec.sparse24_mask(w).to(w.dtype)
    bad = w.clone()
    bad[m - 1, k - 4:k] = torch.tensor([3.0, -3.0, 2.5, 2.0], dtype=BF16, device=dev)  # ONE dense quad (four non-zero codes), in the last tile
    sc, _ = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
    ref = cta.codec.marlin24_compress_w4_full(w, sc, None, group_size=128, group_perm=128 < k // 2)
    torch.cuda.synchronize()
    assert int(ref[3].item()) == 0
    mb = _lib.mailbox(dev.index or 0)
    stream = _lib.stream_of_device(dev)
    # the ticket tree is the caller's (round 6): handed over full of garbage with clear_workspace = 1 once, then left zero by every launch
    tree = torch.full((_lib.Mailbox.M24_VERDICT_WORKSPACE_BYTES // 4,), 0x5a5a5a5a, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for rep in range(30):
        x, want = (bad, 3) if rep % 3 == 1 else (w, 1)
        packed = torch.empty_like(ref[0])
        meta = torch.empty_like(ref[1])
        sp = torch.empty_like(ref[2])
        mb.words[1] = 0
        rc = lib.ct_marlin24_compress_w4_verdict(x.data_ptr(), _lib.BF16, sc.data_ptr(), _lib.BF16, None, -1, m, k, 128, int(128 < k // 2), packed.data_ptr(),
                                                 meta.data_ptr(), sp.data_ptr(), mb.dev + 8, tree.data_ptr(), int(rep == 0), stream)
        assert rc == 0, _lib.last_error()
        got = mb.wait_word(1, 0, stream)
        assert got == want, (rep, got)
        torch.cuda.synchronize()
        assert mb.words[1] == want
        assert int(tree.count_nonzero()) == 0, rep  # every counter back at zero: the next launch needs no clearing
        if want == 1:
            assert torch.equal(packed, ref[0]) and torch.equal(meta, ref[1]) and torch.equal(sp, ref[2])
    # fp32 scales are outside the one-launch kernel: refused, nothing launched, the word untouched
    mb.words[1] = 0
    rc = lib.ct_marlin24_compress_w4_verdict(w.data_ptr(), _lib.BF16, sc.float().data_ptr(), _lib.F32, None, -1, m, k, 128, 0, packed.data_ptr(), meta.data_ptr(),
                                             sp.data_ptr(), mb.dev + 8, tree.data_ptr(), 0, stream)
    assert rc != 0 and mb.words[1] == 0
    # no workspace / a misaligned one: an argument error, nothing launched
    for ws in (None, tree.data_ptr() + 64):
        rc = lib.ct_marlin24_compress_w4_verdict(w.data_ptr(), _lib.BF16, sc.data_ptr(), _lib.BF16, None, -1, m, k, 128, 0, packed.data_ptr(), meta.data_ptr(),
                                                 sp.data_ptr(), mb.dev + 8, ws, 0, stream)
        assert rc == _lib.CT_ERR_INVALID_ARG and "workspace" in _lib.last_error() and mb.words[1] == 0


def _m24_verdict_case(cta, dev, m, k, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    w = torch.randn(m, k, dtype=BF16, device=dev, generator=g)
    w = w * cta.codec.sparse24_mask(w).to(w.dtype)
    bad = w.clone()
    r, c = (seed * 37) % m, ((seed * 101) % (k // 4)) * 4
    bad[r, c:c + 4] = torch.tensor([3.0, -3.0, 2.5, 2.0], dtype=BF16, device=dev)  # ONE dense quad somewhere
    sc, _ = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
    return w, bad, sc


def test_marlin24_verdict_64_launches_in_flight_without_waiting(cta, dev):
    """a plain C caller that pipelines: 64 verdict launches queued back to back on ONE stream, sharing ONE workspace
    (stream order separates them), each with its own pinned word and its own outputs, no wait in between — then 64 words are read: every
    verdict right, every output equal to the waited-for call's.  Then the same 64 over FOUR streams with a workspace per stream, so that
    launches really overlap on the device."""
    import ctypes

    from compressed_tensors_amd import _lib

    lib = _lib.load()
    m, k = 512, 1024
    w, bad, sc = _m24_verdict_case(cta, dev, m, k, 3)
    ref = cta.codec.marlin24_compress_w4_full(w, sc, None, group_size=128, group_perm=128 < k // 2)
    torch.cuda.synchronize()
    h, d = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.ct_mailbox_alloc(8 * 64, ctypes.byref(h), ctypes.byref(d)))
    words = (ctypes.c_int64 * 64).from_address(h.value)
    try:
        for nstreams in (1, 4):
            streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
            trees = [torch.zeros(_lib.Mailbox.M24_VERDICT_WORKSPACE_BYTES // 4, dtype=torch.int32, device=dev) for _ in range(nstreams)]
            outs = [(torch.empty_like(ref[0]), torch.empty_like(ref[1]), torch.empty_like(ref[2])) for _ in range(64)]
            want = [3 if (i * 7) % 5 < 2 else 1 for i in range(64)]
            for i in range(64):
                words[i] = 0
            torch.cuda.synchronize()
            for i in range(64):
                x = bad if want[i] == 3 else w
                st = _lib.stream_on(dev, streams[i % nstreams].cuda_stream)
                rc = lib.ct_marlin24_compress_w4_verdict(x.data_ptr(), _lib.BF16, sc.data_ptr(), _lib.BF16, None, -1, m, k, 128, 1, outs[i][0].data_ptr(),
                                                         outs[i][1].data_ptr(), outs[i][2].data_ptr(), d.value + 8 * i, trees[i % nstreams].data_ptr(), 0, st)
                assert rc == 0, _lib.last_error()
            torch.cuda.synchronize()
            assert [words[i] for i in range(64)] == want, nstreams
            for i in range(64):
                if want[i] == 1:
                    assert all(torch.equal(a, b) for a, b in zip(outs[i], ref[:3])), (nstreams, i)
            assert all(int(t.count_nonzero()) == 0 for t in trees)
    finally:
        lib.ct_mailbox_free(h)


def test_marlin24_verdict_workspaces_on_two_devices(cta, dev):
    """one workspace per device: the same host thread launches on cuda:0 and cuda:1 back to back without waiting (skipped on a one-GPU box)"""
    import ctypes

    from compressed_tensors_amd import _lib

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    lib = _lib.load()
    m, k = 256, 1024
    per_dev = []
    for i in range(2):
        dv = torch.device("cuda", i)
        with torch.cuda.device(dv):
            w, bad, sc = _m24_verdict_case(cta, dv, m, k, 5 + i)
            ref = cta.codec.marlin24_compress_w4_full(w, sc, None, group_size=128, group_perm=True)
            per_dev.append((dv, w, bad, sc, ref, torch.zeros(2080, dtype=torch.int32, device=dv), _lib.mailbox(i)))
            torch.cuda.synchronize(dv)
    for rep in range(10):
        outs = []
        for i, (dv, w, bad, sc, ref, tree, mb) in enumerate(per_dev):
            x = bad if (rep + i) % 2 else w
            o = (torch.empty_like(ref[0]), torch.empty_like(ref[1]), torch.empty_like(ref[2]))
            mb.words[1] = 0
            with torch.cuda.device(dv):
                rc = lib.ct_marlin24_compress_w4_verdict(x.data_ptr(), _lib.BF16, sc.data_ptr(), _lib.BF16, None, -1, m, k, 128, 1, o[0].data_ptr(), o[1].data_ptr(),
                                                         o[2].data_ptr(), mb.dev + 8, tree.data_ptr(), 0, _lib.stream_of_device(dv))
            assert rc == 0, _lib.last_error()
            outs.append(o)
        for i, (dv, w, bad, sc, ref, tree, mb) in enumerate(per_dev):
            assert mb.wait_word(1, 0, _lib.stream_of_device(dv)) == (3 if (rep + i) % 2 else 1)
            torch.cuda.synchronize(dv)
            if (rep + i) % 2 == 0:
                assert all(torch.equal(a, b) for a, b in zip(outs[i], ref[:3]))


def test_marlin24_default_mode_from_32_threads(cta, dev):
    """32 host threads x 50 calls of the raise-from-the-call mode, valid and violating weights mixed (more threads than the sixteen global
    trees the round-5 entry had): each thread owns its mailbox word and its ticket tree, so every verdict is right and every output equal
    to the single-threaded one — the entry is re-entrant as include/ct_hip.h promises (convert_checkpoint's ThreadPoolExecutor,
    entrypoints/convert/convert_checkpoint.py:129-132)."""
    import threading

    M = cta.Marlin24Compressor
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, strategy="group", group_size=128, symmetric=True))
    shapes = [(64, 256), (128, 512), (256, 1024), (512, 512)]
    cases = []
    for i, (m, k) in enumerate(shapes):
        w, bad, sc = _m24_verdict_case(cta, dev, m, k, 11 + i)
        zp = torch.zeros_like(sc, dtype=torch.int8)
        good = {"weight": w, "weight_scale": sc, "weight_zero_point": zp}
        with M.deferred_structure_check():
            ref = M.compress(good, scheme)
        cases.append((good, dict(good, weight=bad), ref))
    torch.cuda.synchronize()
    errors, verdicts = [], [0] * 32

    def run(t):
        try:
            torch.cuda.set_device(dev)
            for r in range(50):
                good, bad, ref = cases[(t + r) % len(cases)]
                if (t * 50 + r) % 3 == 0:
                    with pytest.raises(ValueError, match="2:4 sparsity structure"):
                        M.compress(bad, scheme)
                else:
                    out = M.compress(good, scheme)
                    assert all(torch.equal(out[key], ref[key]) for key in ("weight_packed", "scale_packed", "meta")), (t, r)
                verdicts[t] += 1
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=run, args=(t,)) for t in range(32)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    assert verdicts == [50] * 32


def test_marlin24_default_mode_raises_from_the_call_without_draining_the_stream(cta, dev):
    """the class call in default mode (the ValueError comes from the call itself, as upstream): verdict through the pinned word — same
    outputs as the deferred mode, the error for a dense weight, and again a clean call afterwards; with a kernel queued BEHIND the
    compress on the same stream the call still returns (it does not depend on the stream being empty)"""
    M = cta.Marlin24Compressor
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, strategy="group", group_size=128, symmetric=True))
    torch.manual_seed(5)
    w = torch.randn(1024, 2048, dtype=BF16, device=dev)
    w = w * cta.codec.sparse24_mask(w).to(w.dtype)
    sc, zp = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
    sd = {"weight": w, "weight_scale": sc, "weight_zero_point": zp}
    with M.deferred_structure_check():
        ref = M.compress(sd, scheme)
    for rep in range(20):
        out = M.compress(sd, scheme)
        assert all(torch.equal(out[k], ref[k]) for k in ("weight_packed", "scale_packed", "meta"))
        if rep % 4 == 0:
            dense = dict(sd, weight=torch.randn(1024, 2048, dtype=BF16, device=dev))
            with pytest.raises(ValueError, match="2:4 sparsity structure"):
                M.compress(dense, scheme)


# ----------------------------------------------------------------------------- qparams of the FLOAT schemes
from test_oracle_golden import _qpf_kind  # noqa: E402


@pytest.mark.parametrize("case", cases("qparams_float"), ids=lambda c: c["key"])
def test_qparams_float_golden(golden, cta, dev, case):
    t = golden.case("qparams_float", case["key"])
    gs = t.get("gs")
    if gs is not None:
        got_gs = cta.codec.generate_gparam(d(t["x"], dev))
        assert got_gs.dtype == F32 and eq(got_gs.cpu(), gs)
    s = cta.codec.minmax_qparams_float(d(t["x"], dev), kind=_qpf_kind(case), group_size=case["group_size"], global_scale=d(gs, dev))
    assert eq(s.cpu(), t["scale"])


@pytest.mark.parametrize("xdt", [BF16, F16, F32])
@pytest.mark.parametrize("kind,gsize", [("fp8", None), ("fp8", 128), ("nvfp4", 16), ("mxfp4", 32), ("mxfp8", 32)])
def test_qparams_float_vs_oracle(cta, dev, xdt, kind, gsize):
    """every exponent and significand quarter of the group maximum (the power-of-two rounding of the MX kinds), zeros,
    subnormals, inf and NaN groups, wide rows (the one-wave-per-group kernel) and narrow ones"""
    g = torch.Generator().manual_seed(11)
    for shape in ((64, 256), (3, 8192 * 2), (5, 96) if gsize in (None, 16, 32) else (5, 128)):
        x = (torch.randn(shape, generator=g) * 0.3).to(xdt)
        group = gsize or shape[1]
        ngroups = shape[0] * (shape[1] // group)
        # the maximum of group i sweeps the exponent range and the quarters of the significand
        mags = (2.0 ** torch.linspace(-30, 17, ngroups)) * (1 + 0.25 * (torch.arange(ngroups) % 5))
        xv = x.view(-1, group)
        xv[:, 0] = torch.where(torch.arange(ngroups) % 2 == 0, mags, -mags).to(xdt)
        xv[:, 1:] = (xv[:, 1:].float().clamp(-1, 1) * mags[:, None] * 0.9).to(xdt)
        xv[0] = 0
        if ngroups > 3:
            xv[1, 3] = float("nan")
            xv[2, 2] = float("inf")
            xv[3] = torch.finfo(xdt).tiny / 4 if xdt != F32 else 1e-42
        gs = O.generate_gparam(x[torch.isfinite(x.float()).all(dim=1)]) if kind == "nvfp4" else None
        got = cta.codec.minmax_qparams_float(d(x, dev), kind=kind, group_size=gsize, global_scale=d(gs, dev))
        assert eq(got.cpu(), O.calculate_qparams_float(x, kind=kind, group_size=gsize, global_scale=gs)), shape
    xf = (torch.randn((300, 1000), generator=g) * 7).to(xdt)
    assert eq(cta.codec.generate_gparam(d(xf, dev)).cpu(), O.generate_gparam(xf))


@pytest.mark.parametrize("fmt", ["nvfp4-pack-quantized", "mxfp4-pack-quantized"])
def test_fp4_compress_of_float32_weights(cta, dev, fmt):
    """a float32 checkpoint through the FP4 codecs (declined before round 6): the kernel's IEEE float32 quotient + hardware E2M1 rounding against the
    oracle, special values and quotients on the rounding thresholds included; the class `compress` and `compress_rtn` on float32 weights"""
    g = torch.Generator().manual_seed(5)
    group = 16 if fmt.startswith("nvfp4") else 32
    x = torch.randn((96, 512), generator=g) * 0.7
    sp = torch.tensor([0.0, -0.0, 1e-30, -1e-30, float("inf"), -float("inf"), float("nan"), 1e30, 0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, -0.25, -0.75, -2.5, 6.0, 7.0])
    x.view(-1)[: sp.numel()] = sp
    if group == 16:
        gs = torch.tensor([2.5])
        s = (torch.rand((96, 512 // 16), generator=g) * 2 + 0.05).to(F8).float()
        s[0, 0] = 1.0 * 2.5  # s_eff = 1: the special values above sit exactly on cast_to_fp4's thresholds
    else:
        gs = None
        s = (2.0 ** torch.randint(-6, 3, (96, 512 // 32), generator=g).float()).to(BF16)
        s[0, 0] = 1.0
    got = cta.codec.fp4_quantize_and_pack(d(x, dev), d(s, dev), d(gs, dev), group_size=group)
    ref = O.fp4_compress(x, s, gs, fmt=fmt)
    assert torch.equal(got.cpu(), ref["weight_packed"])
    comp = cta.BaseCompressor.get_value_from_registry(fmt)
    scheme = _fp4_scheme(cta, fmt)
    sd = {"weight": d(x, dev), "weight_scale": d(s, dev)}
    if gs is not None:
        sd["weight_global_scale"] = d(gs, dev)
    out = comp.compress(sd, scheme)
    assert torch.equal(out["weight_packed"].cpu(), ref["weight_packed"]) and torch.equal(out["weight_scale"].cpu().view(torch.uint8), ref["weight_scale"].view(torch.uint8))
    xf = torch.nan_to_num(x, nan=0.0, posinf=3.0, neginf=-3.0).clamp(-9, 9)
    rtn = comp.compress_rtn(d(xf, dev), scheme)
    gs_r = O.generate_gparam(xf) if group == 16 else None
    s_r = O.calculate_qparams_float(xf, kind="nvfp4" if group == 16 else "mxfp4", group_size=group, global_scale=gs_r)
    ref_r = O.fp4_compress(xf, s_r, gs_r, fmt=fmt)
    assert torch.equal(rtn["weight_packed"].cpu(), ref_r["weight_packed"]) and torch.equal(rtn["weight_scale"].cpu().view(torch.uint8), ref_r["weight_scale"].view(torch.uint8))


@pytest.mark.parametrize("fmt", ["nvfp4-pack-quantized", "mxfp4-pack-quantized"])
def test_fp4_module_loops_equal_the_generic_module_path(cta, dev, fmt):
    """NVFP4 / MXFP4PackedCompressor.compress_modules / decompress_modules (one launch per module and direction, parameters rewritten as a delta)
    leave a module exactly as compress_module / decompress_module do: the same entries in the same order, bit-identical tensors, non-trainable
    Parameters, the status — with a bias, a symmetric scheme's zero point, an input global scale and a trainable stray parameter on the module;
    a module the one-launch form declines (float32 weight scale of an MX scheme) takes the generic path inside the loop"""
    from compressed_tensors_amd.quantization import calculate_qparams_from_weight

    comp = cta.BaseCompressor.get_value_from_registry(fmt)
    scheme = _fp4_scheme(cta, fmt)
    g = torch.Generator().manual_seed(77)

    def make(shape, scale_float32=False):
        w = (torch.randn(shape, generator=g) * 0.05).to(BF16).to(dev)
        lin = torch.nn.Linear(shape[1], shape[0], bias=True, device=dev, dtype=BF16)
        lin.weight = torch.nn.Parameter(w, requires_grad=False)
        gs = cta.codec.generate_gparam(w) if fmt.startswith("nvfp4") else None
        scale, zp = calculate_qparams_from_weight(w, scheme.weights, global_scale=gs) if gs is not None else calculate_qparams_from_weight(w, scheme.weights)
        lin.weight_scale = torch.nn.Parameter(scale.float() if scale_float32 else scale, requires_grad=False)
        lin.weight_zero_point = torch.nn.Parameter(zp, requires_grad=False)
        if gs is not None:
            lin.weight_global_scale = torch.nn.Parameter(gs, requires_grad=False)
        lin.input_global_scale = torch.nn.Parameter(torch.ones(1, device=dev), requires_grad=False)
        lin.stray = torch.nn.Parameter(torch.ones(3, device=dev), requires_grad=True)
        lin.quantization_scheme = scheme
        return lin

    def same(a, b):
        assert list(a._parameters) == list(b._parameters) and list(a._buffers) == list(b._buffers)
        for k in a._parameters:
            x, y = a._parameters[k], b._parameters[k]
            assert type(x) is type(y) and x.requires_grad == y.requires_grad and x.dtype == y.dtype and x.shape == y.shape, k
            assert torch.equal(x.data.view(torch.uint8) if x.dtype == F8 else x.data, y.data.view(torch.uint8) if y.dtype == F8 else y.data), k
        assert a.quantization_status == b.quantization_status

    import copy
    for shape, f32 in (((64, 512), False), ((48, 96), False), ((64, 512), True)):
        if f32 and fmt.startswith("nvfp4"):
            continue  # NVFP4 scales ARE float32
        a = make(shape, f32)
        b = copy.deepcopy(a)
        b.quantization_scheme = scheme
        comp.compress_modules([a])
        comp.compress_module(b)
        same(a, b)
        assert "weight" not in a._parameters and "weight_zero_point" not in a._parameters and a._parameters["weight_packed"].dtype == torch.uint8
        comp.decompress_modules([a])
        comp.decompress_module(b)
        same(a, b)
        assert a._parameters["weight"].dtype == BF16 and a._parameters["weight_scale"].dtype == BF16


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_fp4_scales_written_by_the_codec_launches(cta, dev, xdt):
    """round 6: the compressors' scale tensors come out of the weight launches — the STORED scale on the way in (NVFP4 `scale.to(float8_e4m3fn)` with
    torch's overflow-to-NaN rule at every boundary; MXFP4 `compress_mx_scale` for EVERY 16-bit scale pattern, through the code table), the bfloat16 scale
    on the way back (all 256 stored bytes of both kinds) — each against the reference's own expression evaluated on the CPU"""
    g = torch.Generator().manual_seed(23)
    # NVFP4: float32 scales, 64 x 256 of them
    rows, cols = 64, 16 * 256
    w = (torch.randn((rows, cols), generator=g) * 0.5).to(xdt)
    edge = torch.tensor([0.0, -0.0, 2.0 ** -9, 2.0 ** -10, 2.0 ** -10 * 1.0001, 2.0 ** -11, 3 * 2.0 ** -10, 1e-30, 447.9, 448.0, 455.9, 464.0, 464.00003, 479.9, 480.0, 1e9, float("inf"),
                         -float("inf"), float("nan"), -448.0, -464.0, -464.1, -1e-3, 0.0625, 0.0703125, 17.0, 18.0, 19.0, 1.0625, 1.1875, 240.0, 248.0, 232.0])
    sc = torch.exp(torch.randn((rows, cols // 16), generator=g) * 4.0) * torch.where(torch.rand((rows, cols // 16), generator=g) < 0.1, -1.0, 1.0)
    sc.view(-1)[: edge.numel()] = edge
    gs = torch.tensor([3.0])
    got = cta.codec.fp4_quantize_and_pack_stored(d(w, dev), d(sc, dev), d(gs, dev), group_size=16, scale_dtype=F8)
    assert got is not None
    assert eq_f8(got[1].cpu(), sc.to(F8)) and torch.equal(got[0].cpu(), cta.codec.fp4_quantize_and_pack(d(w, dev), d(sc, dev), d(gs, dev), group_size=16).cpu())
    assert torch.equal(got[1].cpu().view(torch.uint8)[sc == sc], sc.to(F8).view(torch.uint8)[sc == sc])  # the sign of a zero / of an overflow included
    # MXFP4: every pattern of the scale dtype exactly once
    rows, cols = 64, 32 * 1024
    w = (torch.randn((rows, cols), generator=g) * 0.5).to(xdt)
    every = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(xdt).reshape(rows, cols // 32)
    got = cta.codec.fp4_quantize_and_pack_stored(d(w, dev), d(every, dev), None, group_size=32, scale_dtype=torch.uint8)
    assert got is not None
    ref_codes = (127 + torch.floor(torch.log2(every)).to(torch.int32)).to(torch.uint8)  # mx_utils.py:18-31
    assert torch.equal(got[1].cpu(), ref_codes) and torch.equal(got[0].cpu(), cta.codec.fp4_quantize_and_pack(d(w, dev), d(every, dev), None, group_size=32).cpu())
    # the helpers by themselves (the MXFP8 codec's scale conversions): one launch each, the same codes
    assert torch.equal(cta.codec.compress_mx_scale(d(every, dev)).cpu(), ref_codes)
    all_codes = torch.arange(256, dtype=torch.uint8).reshape(4, 64)
    assert eq(cta.codec.decompress_mx_scale(d(all_codes, dev)).cpu(), 2.0 ** (all_codes.to(torch.int32) - 127).to(BF16))
    # layouts outside the kernel are declined (the class then converts the scale itself)
    assert cta.codec.fp4_quantize_and_pack_stored(d(w, dev), d(every.float(), dev), None, group_size=32, scale_dtype=torch.uint8) is None
    assert cta.codec.fp4_quantize_and_pack_stored(d(w, dev), d(every, dev), None, group_size=32, scale_dtype=torch.int16) is None
    # the way back: all 256 stored bytes, both kinds
    codes = torch.arange(256, dtype=torch.uint8).repeat(8).reshape(8, 256)
    packed = torch.randint(0, 256, (8, 256 * 16), generator=g, dtype=torch.uint8)
    wv, sv = cta.codec.fp4_unpack_and_dequantize(d(packed[:, : 256 * 8].contiguous(), dev), d(codes.view(F8), dev), d(gs, dev), group_size=16, scale_kind="f8e4m3", return_scale=True)
    assert sv.dtype == BF16 and eq(sv.cpu(), codes.view(F8).to(BF16))
    assert eq(wv.cpu(), cta.codec.fp4_unpack_and_dequantize(d(packed[:, : 256 * 8].contiguous(), dev), d(codes.view(F8), dev), d(gs, dev), group_size=16, scale_kind="f8e4m3").cpu())
    wv, sv = cta.codec.fp4_unpack_and_dequantize(d(packed, dev), d(codes, dev), None, group_size=32, scale_kind="e8m0", return_scale=True)
    assert sv.dtype == BF16 and eq(sv.cpu(), 2.0 ** (codes.to(torch.int32) - 127).to(BF16))  # mx_utils.py:34-44
    assert eq(wv.cpu(), cta.codec.fp4_unpack_and_dequantize(d(packed, dev), d(codes, dev), None, group_size=32, scale_kind="e8m0").cpu())


@pytest.mark.parametrize("xdt", [BF16, F16, F32])
def test_generate_gparam_on_the_device(cta, dev, xdt):
    """ct_generate_gparam (row maxima + one finishing workgroup) against the oracle's restatement of helpers.py:308-337: ordinary weights of several
    shapes (wide rows, many rows, 3-D), all zeros (amax clamps to tiny, the quotient overflows -> 1), a NaN, an inf, the dtype's largest value,
    subnormal-only weights, a negative maximum"""
    g = torch.Generator().manual_seed(41)
    cases = [torch.randn((300, 1000), generator=g) * 7, torch.randn((5, 8192 * 3), generator=g) * 0.02, torch.randn((2100, 64), generator=g), torch.randn((4, 16, 96), generator=g),
             torch.zeros((16, 64)), -torch.rand((8, 128), generator=g) - 3.0, torch.full((8, 64), 1e-40), torch.randn((1, 8), generator=g)]
    nan = torch.randn((64, 256), generator=g); nan[17, 5] = float("nan")
    inf = torch.randn((64, 256), generator=g); inf[63, 255] = -float("inf")
    big = torch.randn((64, 256), generator=g); big[0, 0] = torch.finfo(xdt).max
    for x in cases + [nan, inf, big]:
        x = x.to(xdt)
        got = cta.codec.generate_gparam(d(x, dev))
        assert got.dtype == F32 and got.shape == (1,) and eq(got.cpu(), O.generate_gparam(x)), (tuple(x.shape), float(got))
    with pytest.raises(ValueError):
        cta.codec.generate_gparam(torch.empty((0, 64), dtype=xdt, device=dev))


def test_float_schemes_end_to_end_from_the_dense_weight(cta, dev):
    """observer -> calculate_qparams -> compress -> decompress for every FLOAT scheme, all on the device, against the
    oracle's composition of the same steps"""
    from compressed_tensors_amd.quantization import calculate_qparams_from_weight

    torch.manual_seed(0)
    w = (torch.randn(64, 512) * 0.04).to(BF16)
    wd = w.to(dev)
    for fmt, args, kind, group in (
        ("nvfp4-pack-quantized", _fp4_scheme(cta, "nvfp4-pack-quantized").weights, "nvfp4", 16),
        ("mxfp4-pack-quantized", _fp4_scheme(cta, "mxfp4-pack-quantized").weights, "mxfp4", 32),
        ("mxfp8-quantized", cta.QuantizationArgs(num_bits=8, type="float", strategy="group", group_size=32, scale_dtype=torch.uint8, zp_dtype=torch.uint8), "mxfp8", 32),
        ("float-quantized", cta.QuantizationArgs(num_bits=8, type="float", strategy="channel"), "fp8", None),
    ):
        gs = cta.codec.generate_gparam(wd) if kind == "nvfp4" else None
        scale, zp = calculate_qparams_from_weight(wd, args, global_scale=gs)
        gs_ref = O.generate_gparam(w) if kind == "nvfp4" else None
        s_ref = O.calculate_qparams_float(w, kind=kind, group_size=group, global_scale=gs_ref)
        assert eq(scale.cpu(), s_ref) and not bool(zp.view(torch.uint8).any()), fmt
        act = cta.QuantizationArgs(num_bits=8, type="float", strategy="tensor") if fmt == "float-quantized" else None
        scheme = cta.QuantizationScheme(targets=["Linear"], weights=args, input_activations=act)
        comp = cta.BaseCompressor.get_value_from_registry(fmt)
        sd = {"weight": wd, "weight_scale": scale, "weight_zero_point": zp}
        if gs is not None:
            sd["weight_global_scale"] = gs
        back = comp.decompress(comp.compress(sd, scheme), scheme)["weight"]
        if kind in ("nvfp4", "mxfp4"):
            ref = O.fp4_decompress(O.fp4_compress(w, s_ref, gs_ref, fmt=fmt), fmt=fmt)["weight"]
        else:
            q = O.quantize(w, s_ref, None, num_bits=8, strategy="group" if group else "channel", group_size=group, dtype=F8, qtype="float")
            sdec = O.e8m0_decode(O.e8m0_encode(s_ref)) if kind == "mxfp8" else s_ref
            ref = O.dequantize(q, sdec, None)
        assert eq(back.cpu(), ref), fmt
        assert float((back.float().cpu() - w.float()).abs().max()) < (0.05 if kind in ("nvfp4", "mxfp4") else 0.02)


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_rtn_mxfp4_one_pass(cta, dev, xdt):
    """ct_rtn_mxfp4_quant_pack == observer (MX branch of calculate_qparams) -> fp4 compress -> E8M0 encoding, against the
    oracle and against the kernels it fuses; group maxima sweep every exponent and significand quarter"""
    g = torch.Generator().manual_seed(21)
    for shape in ((64, 256), (7, 96), (1, 32), (33, 4096)):
        x = (torch.randn(shape, generator=g) * 0.3).to(xdt)
        ngroups = x.numel() // 32
        lo, hi = (-30, 17) if xdt == BF16 else (-20, 14)
        mags = (2.0 ** torch.linspace(lo, hi, ngroups)) * (1 + 0.25 * (torch.arange(ngroups) % 5))
        xv = x.view(-1, 32)
        xv[:, 0] = torch.where(torch.arange(ngroups) % 2 == 0, mags, -mags).to(xdt)
        xv[:, 1:] = (xv[:, 1:].float().clamp(-1, 1) * mags[:, None] * 0.9).to(xdt)
        xv[0] = 0
        if ngroups > 2:
            xv[1] = torch.finfo(xdt).tiny / 4
        packed, code, scale = cta.codec.rtn_mxfp4_quantize_and_pack(d(x, dev), return_scale=True)
        s_ref = O.calculate_qparams_float(x, kind="mxfp4", group_size=32)
        ref = O.fp4_compress(x, s_ref, None, fmt="mxfp4-pack-quantized")
        assert eq(scale.cpu(), s_ref), shape
        assert torch.equal(code.cpu(), ref["weight_scale"]) and torch.equal(packed.cpu(), ref["weight_packed"]), shape
        s2 = cta.codec.minmax_qparams_float(d(x, dev), kind="mxfp4", group_size=32)
        assert torch.equal(packed, cta.codec.fp4_quantize_and_pack(d(x, dev), s2, None, group_size=32))
        assert torch.equal(code, cta.codec.compress_mx_scale(s2))
    scheme = _fp4_scheme(cta, "mxfp4-pack-quantized")
    w = torch.randn((128, 512), generator=g).to(xdt).to(dev)
    got = cta.MXFP4PackedCompressor.compress_rtn(w, scheme)
    s2 = cta.codec.minmax_qparams_float(w, kind="mxfp4", group_size=32)
    ref = cta.MXFP4PackedCompressor.compress({"weight": w, "weight_scale": s2}, scheme)
    assert sorted(got) == sorted(ref) and all(torch.equal(got[k], ref[k]) for k in ref)
    back = cta.MXFP4PackedCompressor.decompress(got, scheme)["weight"]
    assert float((back.float() - w.float()).abs().max()) <= float(w.float().abs().max()) * 0.26


@pytest.mark.parametrize("xdt", [BF16, F16])
def test_rtn_nvfp4_one_pass(cta, dev, xdt):
    """ct_rtn_nvfp4_quant_pack == generate_gparam -> observer (float8 scales under the global scale) -> fp4 compress"""
    g = torch.Generator().manual_seed(23)
    for shape in ((64, 256), (7, 96), (1, 32), (33, 4096)):
        x = (torch.randn(shape, generator=g) * 0.3).to(xdt)
        ngroups = x.numel() // 16
        mags = (2.0 ** torch.linspace(-12, 3, ngroups)) * (1 + 0.25 * (torch.arange(ngroups) % 5))
        xv = x.view(-1, 16)
        xv[:, 0] = torch.where(torch.arange(ngroups) % 2 == 0, mags, -mags).to(xdt)
        xv[:, 1:] = (xv[:, 1:].float().clamp(-1, 1) * mags[:, None] * 0.9).to(xdt)
        xv[0] = 0
        packed, s8, gs, scale = cta.codec.rtn_nvfp4_quantize_and_pack(d(x, dev), return_scale=True)
        gs_ref = O.generate_gparam(x)
        s_ref = O.calculate_qparams_float(x, kind="nvfp4", group_size=16, global_scale=gs_ref)
        ref = O.fp4_compress(x, s_ref, gs_ref, fmt="nvfp4-pack-quantized")
        assert eq(gs.cpu(), gs_ref) and eq(scale.cpu(), s_ref), shape
        assert torch.equal(s8.cpu().view(torch.uint8), ref["weight_scale"].view(torch.uint8)) and torch.equal(packed.cpu(), ref["weight_packed"]), shape
    scheme = _fp4_scheme(cta, "nvfp4-pack-quantized")
    w = torch.randn((128, 512), generator=g).to(xdt).to(dev)
    got = cta.NVFP4PackedCompressor.compress_rtn(w, scheme)
    gs2 = cta.codec.generate_gparam(w)
    s2 = cta.codec.minmax_qparams_float(w, kind="nvfp4", group_size=16, global_scale=gs2)
    ref = cta.NVFP4PackedCompressor.compress({"weight": w, "weight_scale": s2, "weight_global_scale": gs2}, scheme)
    assert sorted(got) == sorted(ref)
    for k in ref:
        assert torch.equal(got[k].view(torch.uint8) if got[k].dtype == F8 else got[k], ref[k].view(torch.uint8) if ref[k].dtype == F8 else ref[k]), k
    back = cta.NVFP4PackedCompressor.decompress(got, scheme)["weight"]
    assert float((back.float() - w.float()).abs().max()) <= float(w.float().abs().max()) * 0.26


@pytest.mark.parametrize("xdt", [BF16, F16])
@pytest.mark.parametrize("qtype,symmetric", [("int", True), ("int", False), ("float", True)])
def test_rtn_channel8_one_pass(cta, dev, xdt, qtype, symmetric):
    """ct_rtn_quant_channel8 == per-row observer + calculate_qparams + quantize (int8 / float8), for every row width class"""
    g = torch.Generator().manual_seed(29)
    for shape in ((16, 8), (9, 2048), (5, 2056), (7, 4096), (3, 11008), (4, 16384), (300, 512)):
        x = (torch.randn(shape, generator=g) * torch.logspace(-3, 2, shape[0]).reshape(-1, 1)).to(xdt)
        sp = special_values(xdt)
        sp = sp[torch.isfinite(sp.float())]
        n = min(sp.numel(), shape[1])
        x[0, :n] = sp[:n]
        if shape[0] > 2:
            x[1] = 0
            x[2] = x[2].abs() + 0.01
        q, scale, zp = cta.codec.rtn_quantize_channel8(d(x, dev), qtype=qtype, symmetric=symmetric)
        if qtype == "int":
            s_ref, z_ref = O.calculate_qparams_minmax(x, num_bits=8, group_size=None, symmetric=symmetric)
            q_ref = O.quantize(x, s_ref, z_ref, num_bits=8, strategy="channel", dtype=torch.int8)
            assert eq(scale.cpu(), s_ref) and torch.equal(zp.cpu(), z_ref) and torch.equal(q.cpu(), q_ref), shape
        else:
            s_ref = O.calculate_qparams_float(x, kind="fp8")
            q_ref = O.quantize(x, s_ref, torch.zeros_like(s_ref, dtype=F8), num_bits=8, strategy="channel", dtype=F8, qtype="float")
            assert eq(scale.cpu(), s_ref) and eq_f8(q.cpu(), q_ref) and not bool(zp.view(torch.uint8).any()), shape
            s2 = cta.codec.minmax_qparams_float(d(x, dev), kind="fp8")
            q2 = cta.codec.quantize_tensor(d(x, dev), s2, torch.zeros_like(s2, dtype=F8), num_bits=8, strategy="channel", dtype=F8, qtype="float")
            assert torch.equal(q.view(torch.uint8), q2.view(torch.uint8))


def test_naive_compress_rtn_matches_observer_plus_compress(cta, dev):
    from compressed_tensors_amd.quantization import calculate_qparams_from_weight

    torch.manual_seed(5)
    w = torch.randn((96, 1024), dtype=BF16, device=dev)
    act = cta.QuantizationArgs(num_bits=8, type="float", strategy="tensor")
    for fmt, wargs in (("float-quantized", cta.QuantizationArgs(num_bits=8, type="float", strategy="channel")),
                       ("int-quantized", cta.QuantizationArgs(num_bits=8, strategy="channel", symmetric=True)),
                       ("int-quantized", cta.QuantizationArgs(num_bits=8, strategy="channel", symmetric=False)),
                       ("naive-quantized", cta.QuantizationArgs(num_bits=8, strategy="group", group_size=128, symmetric=True))):
        scheme = cta.QuantizationScheme(targets=["Linear"], weights=wargs, input_activations=act)
        comp = cta.BaseCompressor.get_value_from_registry(fmt)
        got = comp.compress_rtn(w, scheme)
        scale, zp = calculate_qparams_from_weight(w, wargs)
        ref = comp.compress({"weight": w, "weight_scale": scale, "weight_zero_point": zp}, scheme)
        assert sorted(got) == sorted(ref), fmt
        for k in ref:
            a, b = got[k], ref[k]
            assert torch.equal(a.view(torch.uint8) if a.dtype == F8 else a, b.view(torch.uint8) if b.dtype == F8 else b), (fmt, k)


def test_model_compressor_rtn_mixed_schemes(cta, dev):
    """compress_model_rtn: a model with W4A16, MXFP4, NVFP4 and FP8 modules compressed straight from the dense weights; the first
    forward decompresses; every weight equals the oracle's observer -> compress -> decompress of that scheme"""
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 512, bias=True), torch.nn.Linear(512, 256, bias=False), torch.nn.Linear(256, 384, bias=False),
                                torch.nn.Linear(384, 128, bias=False)).to(dev).to(BF16)
    act = cta.QuantizationArgs(num_bits=8, type="float", strategy="tensor")
    schemes = [cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True)),
               _fp4_scheme(cta, "mxfp4-pack-quantized"), _fp4_scheme(cta, "nvfp4-pack-quantized"),
               cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=8, type="float", strategy="channel"), input_activations=act)]
    expect = []
    for m, scheme in zip(model, schemes):
        m.quantization_scheme = scheme
        w = m.weight.data.cpu()
        wa = scheme.weights
        if wa.type.value == "int":
            s, z = O.calculate_qparams_minmax(w, num_bits=4, group_size=128, symmetric=True)
            expect.append(O.fake_quantize(w, s, z, num_bits=4, strategy="group", group_size=128))
        elif wa.num_bits == 4:
            fmt = "mxfp4-pack-quantized" if wa.group_size == 32 else "nvfp4-pack-quantized"
            gs = O.generate_gparam(w) if wa.group_size == 16 else None
            s = O.calculate_qparams_float(w, kind="mxfp4" if gs is None else "nvfp4", group_size=wa.group_size, global_scale=gs)
            expect.append(O.fp4_decompress(O.fp4_compress(w, s, gs, fmt=fmt), fmt=fmt)["weight"])
        else:
            s = O.calculate_qparams_float(w, kind="fp8")
            q = O.quantize(w, s, torch.zeros_like(s, dtype=F8), num_bits=8, strategy="channel", dtype=F8, qtype="float")
            expect.append(O.dequantize(q, s, None))
    bias0 = model[0].bias.data.clone()
    cta.ModelCompressor().compress_model_rtn(model)
    assert [m.quantization_scheme.format.value for m in model] == ["pack-quantized", "mxfp4-pack-quantized", "nvfp4-pack-quantized", "float-quantized"]
    assert model[0].weight_packed.dtype == torch.int32 and model[1].weight_scale.dtype == torch.uint8 and model[2].weight_scale.dtype == F8
    assert model[3].weight.dtype == F8 and torch.equal(model[0].bias.data, bias0)
    y = model(torch.randn(4, 256, device=dev, dtype=BF16))
    assert y.shape == (4, 128)
    for m, ref in zip(model, expect):
        assert torch.equal(m.weight.data.cpu(), ref)
