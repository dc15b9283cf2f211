"""VERDICT r04 missing #4: under `install(patch_functions=True)` the reference's PLAIN functions — `dequantize` / `fake_quantize`
(/root/reference/src/compressed_tensors/quantization/lifecycle/forward.py:76-181) and `pack_to_int32` / `unpack_from_int32`
(compressors/pack_quantized/helpers.py:20-180, bound by name at pack_quantized/base.py:11-14) — called on GPU tensors must reach the
HIP kernels (launch counts through the C ABI > 0) and return, bit for bit, what the same upstream call returns on CPU tensors."""
import collections
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_import.available(), reason="no reference on this machine: run oracle/stage_ref.py in the build container")]

COUNTED = ("ct_dequantize", "ct_fake_quantize", "ct_quantize", "ct_pack_int32", "ct_unpack_int32", "ct_pack_int32_dim0", "ct_unpack_int32_dim0")


@pytest.fixture()
def patched():
    ref_import.import_reference()
    import compressed_tensors.compressors.pack_quantized.base as base_mod
    import compressed_tensors.compressors.pack_quantized.helpers as helpers_mod
    import compressed_tensors.quantization.lifecycle.forward as forward_mod
    import compressed_tensors_amd.install as ct_amd
    from compressed_tensors_amd import _lib

    originals = types.SimpleNamespace(pack_to_int32=helpers_mod.pack_to_int32, unpack_from_int32=helpers_mod.unpack_from_int32,
                                      dequantize=forward_mod.dequantize, fake_quantize=forward_mod.fake_quantize, quantize=forward_mod.quantize)
    assert not hasattr(originals.dequantize, "_ct_original"), "a previous test left the functions patched"
    ct_amd.install(patch_functions=True)
    lib = _lib.load()
    counts = collections.Counter()
    saved = {n: getattr(lib, n) for n in COUNTED}
    for n in COUNTED:
        def counted(*a, _o=saved[n], _n=n):
            counts[_n] += 1
            return _o(*a)
        setattr(lib, n, counted)
    try:
        yield types.SimpleNamespace(helpers=helpers_mod, base=base_mod, forward=forward_mod, originals=originals, counts=counts, ct_amd=ct_amd)
    finally:
        for n in COUNTED:
            setattr(lib, n, saved[n])
        ct_amd.uninstall()


def _qparams(w, args):
    from compressed_tensors.quantization.utils import calculate_qparams

    st = args.strategy.value if hasattr(args.strategy, "value") else args.strategy
    if st == "group":
        x = w.unflatten(-1, (-1, args.group_size))
        mn, mx = x.amin(-1), x.amax(-1)
    elif st == "channel":
        mn, mx = w.amin(-1, keepdim=True), w.amax(-1, keepdim=True)
    else:
        mn, mx = w.amin().reshape(1), w.amax().reshape(1)
    return calculate_qparams(mn, mx, args)


def _bits(t):
    t = t.detach().cpu().contiguous()
    return t.view(torch.uint8) if t.dtype != torch.bool else t


ARGS = [dict(num_bits=b, strategy=st, symmetric=sym, **({"group_size": 128} if st == "group" else {}))
        for b in (4, 8) for st in ("group", "channel", "tensor") for sym in (True, False)]


@pytest.mark.parametrize("kw", ARGS, ids=[f"w{k['num_bits']}_{k['strategy']}_{'sym' if k['symmetric'] else 'asym'}" for k in ARGS])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_patched_dequantize_and_fake_quantize_reach_the_kernels(patched, kw, dtype):
    from compressed_tensors.quantization import QuantizationArgs

    p, dev = patched, torch.device("cuda:0")
    args = QuantizationArgs(**kw)
    torch.manual_seed(kw["num_bits"] * 7 + len(kw["strategy"]))
    x = torch.randn(192, 512, dtype=torch.float32).mul_(0.07).to(dtype)
    scale, zp = _qparams(x, args)
    scale = scale.to(dtype)
    x_q = p.originals.quantize(x, scale, zp, args, dtype=torch.int8)  # upstream, CPU: the int8 codes a compressor stores (naive_quantized/base.py:100-108)
    ref_dq = p.originals.dequantize(x_q, scale, zp, args=args)
    ref_dq_inferred = p.originals.dequantize(x_q, scale, zp)  # strategy inferred from the scale's shape (forward.py:99-130)
    ref_fq = p.originals.fake_quantize(x, scale, zp, args)

    assert p.forward.dequantize is not p.originals.dequantize and p.forward.dequantize._ct_original is p.originals.dequantize
    before = dict(p.counts)
    got_dq = p.forward.dequantize(x_q.to(dev), scale.to(dev), zp.to(dev), args=args)
    assert p.counts["ct_dequantize"] == before.get("ct_dequantize", 0) + 1, dict(p.counts)
    got_dq_inferred = p.forward.dequantize(x_q.to(dev), scale.to(dev), zp.to(dev))
    assert p.counts["ct_dequantize"] == before.get("ct_dequantize", 0) + 2, dict(p.counts)
    got_fq = p.forward.fake_quantize(x.to(dev), scale.to(dev), zp.to(dev), args)
    assert p.counts["ct_fake_quantize"] == before.get("ct_fake_quantize", 0) + 1, dict(p.counts)
    for got, ref, what in ((got_dq, ref_dq, "dequantize"), (got_dq_inferred, ref_dq_inferred, "dequantize (inferred)"), (got_fq, ref_fq, "fake_quantize")):
        assert got.is_cuda and got.dtype == ref.dtype and got.shape == ref.shape, what
        assert torch.equal(_bits(got), _bits(ref)), what
    # CPU tensors still reach upstream's own body: no launch
    n = sum(p.counts.values())
    assert torch.equal(_bits(p.forward.fake_quantize(x, scale, zp, args)), _bits(ref_fq)) and sum(p.counts.values()) == n


@pytest.mark.parametrize("bits", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("shape,packed_dim", [((128, 512), 1), ((96, 520), 1), ((130, 256), 0), ((3, 64, 136), 1)])
def test_patched_pack_and_unpack_reach_the_kernels(patched, bits, shape, packed_dim):
    p, dev = patched, torch.device("cuda:0")
    torch.manual_seed(bits * 31 + shape[-1])
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1))
    q = torch.randint(lo, hi, shape, dtype=torch.int8)
    ref = p.originals.pack_to_int32(q, bits, packed_dim=packed_dim)
    ref_back = p.originals.unpack_from_int32(ref, bits, torch.Size(shape), packed_dim=packed_dim)
    assert torch.equal(ref_back, q)
    sfx = "_dim0" if packed_dim == 0 else ""  # packing along rows (the zero points of pack_quantized/base.py:107) has its own entry
    for mod in (p.helpers, p.base):  # base.py:11-14 binds the two names at import time: both bindings are the wrapper
        before = dict(p.counts)
        got = mod.pack_to_int32(q.to(dev), bits, packed_dim=packed_dim)
        assert p.counts["ct_pack_int32" + sfx] == before.get("ct_pack_int32" + sfx, 0) + 1, dict(p.counts)
        assert got.is_cuda and got.dtype == ref.dtype and got.shape == ref.shape
        assert torch.equal(got.cpu(), ref)
        back = mod.unpack_from_int32(got, bits, torch.Size(shape), packed_dim=packed_dim)
        assert p.counts["ct_unpack_int32" + sfx] == before.get("ct_unpack_int32" + sfx, 0) + 1, dict(p.counts)
        assert back.is_cuda and back.dtype == torch.int8 and torch.equal(back.cpu(), q)


def test_a_module_imported_after_the_first_install_is_covered_by_the_next(patched):
    """install.py: `_patch_functions` re-scans sys.modules on every install(patch_functions=True) (round 4 ran it once only)"""
    p = patched
    late = types.ModuleType("compressed_tensors._imported_late")
    late.dequantize = p.originals.dequantize  # `from ...forward import dequantize` executed before install() would look like this
    late.pack_to_int32 = p.originals.pack_to_int32
    sys.modules[late.__name__] = late
    try:
        p.ct_amd.install(patch_functions=True)
        assert late.dequantize is p.forward.dequantize and late.pack_to_int32 is p.helpers.pack_to_int32
        p.ct_amd.uninstall()
        assert late.dequantize is p.originals.dequantize and late.pack_to_int32 is p.originals.pack_to_int32
        assert p.forward.dequantize is p.originals.dequantize
    finally:
        del sys.modules[late.__name__]
