"""The drop-in route of INTEGRATION.md §A against the REAL upstream package (CPU-only here: the build
container has the reference but no GPU, the GPU box has a GPU but no reference — so this checks the
plug-in plumbing, not the kernels): install() swaps the registry entries with subclasses of the
upstream codecs, registers the `_quantize` ImplBackend backend, hands CPU tensors to the upstream
implementation unchanged, and uninstall() restores the registry."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="upstream reference sources not present on this machine")


@pytest.fixture()
def upstream():
    ct = ref_import.import_reference()
    import compressed_tensors_amd.install as ct_amd

    yield ct, ct_amd
    ct_amd.uninstall()


def test_install_swaps_registry_and_falls_through_on_cpu(upstream):
    ct, ct_amd = upstream
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme
    from compressed_tensors.utils.impl_backend import ImplBackend

    before = {f: BaseCompressor.get_value_from_registry(f) for f in ("pack-quantized", "naive-quantized", "int-quantized")}
    ct_amd.install()
    ct_amd.install()  # idempotent
    after = {f: BaseCompressor.get_value_from_registry(f) for f in before}
    for f in before:
        assert after[f] is not before[f] and issubclass(after[f], before[f]), f
        assert after[f].__name__.endswith("MI355X")
    assert "_quantize_mi355x" in ImplBackend._fn_registry

    # CPU tensors: the subclass defers to upstream, results identical to upstream's own
    torch.manual_seed(0)
    w = torch.randn(32, 256, dtype=torch.bfloat16)
    args = QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = QuantizationScheme(targets=["Linear"], weights=args)
    amax = w.reshape(32, 2, 128).abs().amax(dim=-1).float()
    scale = (amax / 7.5).to(torch.bfloat16)
    sd = {"weight": w, "weight_scale": scale, "weight_zero_point": torch.zeros(32, 2, dtype=torch.int8)}
    ref = before["pack-quantized"].compress(sd, scheme)
    got = after["pack-quantized"].compress(sd, scheme)
    assert ref.keys() == got.keys()
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    back_ref = before["pack-quantized"].decompress(ref, scheme)["weight"]
    back_got = after["pack-quantized"].decompress(got, scheme)["weight"]
    assert torch.equal(back_ref, back_got)

    # the upstream module-level entry points resolve to the swapped classes (lookup by format string at call time)
    lin = torch.nn.Linear(256, 32, bias=False).to(torch.bfloat16)
    lin.weight.data.copy_(w)
    lin.quantization_scheme = scheme
    lin.register_parameter("weight_scale", torch.nn.Parameter(scale, requires_grad=False))
    lin.register_parameter("weight_zero_point", torch.nn.Parameter(torch.zeros(32, 2, dtype=torch.int8), requires_grad=False))
    from compressed_tensors.compressors import compress_module, decompress_module

    compress_module(lin)
    assert torch.equal(lin.weight_packed.data, ref["weight_packed"])
    decompress_module(lin)
    assert torch.equal(lin.weight.data, back_ref)

    ct_amd.uninstall()
    for f in before:
        assert BaseCompressor.get_value_from_registry(f) is before[f]


def test_amd_scheme_objects_are_duck_compatible_with_upstream(upstream):
    """our host package consumes upstream's pydantic QuantizationArgs / QuantizationScheme unchanged"""
    ct, _ = upstream
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme

    import compressed_tensors_amd as cta

    args = QuantizationArgs(num_bits=4, group_size=128, symmetric=False, strategy="group")
    scheme = QuantizationScheme(targets=["Linear"], weights=args)
    assert cta.PackedQuantizationCompressor.compression_param_names(scheme) == \
        ct.compressors.BaseCompressor.get_value_from_registry("pack-quantized").compression_param_names(scheme)
    assert cta.PackedQuantizationCompressor.can_compress(torch.nn.Linear, scheme)
    # meta tensors take the shape-only path without touching the GPU
    sd = {"weight": torch.empty(64, 256, dtype=torch.bfloat16, device="meta"), "weight_scale": torch.empty(64, 2, dtype=torch.bfloat16, device="meta"),
          "weight_zero_point": torch.empty(64, 2, dtype=torch.int8, device="meta")}
    out = cta.PackedQuantizationCompressor.compress(sd, scheme)
    ref = ct.compressors.BaseCompressor.get_value_from_registry("pack-quantized").compress(sd, scheme)
    assert out.keys() == ref.keys()
    for k in out:
        assert out[k].shape == ref[k].shape and out[k].dtype == ref[k].dtype and out[k].device.type == ref[k].device.type, k


def test_install_covers_the_fp4_codecs(upstream):
    """nvfp4 / mxfp4 registry entries are swapped too and the FP4 primitives get ImplBackend backends; CPU inputs
    still take upstream's own code"""
    ct, ct_amd = upstream
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.compressors.nvfp4.helpers import pack_fp4_to_uint8
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme
    from compressed_tensors.utils.impl_backend import ImplBackend

    fmts = ("nvfp4-pack-quantized", "mxfp4-pack-quantized")
    before = {f: BaseCompressor.get_value_from_registry(f) for f in fmts}
    ct_amd.install()
    for f in fmts:
        after = BaseCompressor.get_value_from_registry(f)
        assert after is not before[f] and issubclass(after, before[f])
    assert {"pack_fp4_to_uint8_mi355x", "cast_to_fp4_mi355x"} <= set(ImplBackend._fn_registry)
    assert [fn.__name__ for fn, _, _ in ImplBackend._backends["pack_fp4_to_uint8"]] == ["pack_fp4_to_uint8_mi355x"]

    torch.manual_seed(0)
    w = torch.randn(8, 64, dtype=torch.bfloat16)
    s = torch.exp2(torch.floor(torch.log2(w.float().reshape(8, 2, 32).abs().amax(-1))) - 2).to(torch.bfloat16)
    args = QuantizationArgs(num_bits=4, type="float", strategy="group", group_size=32, symmetric=True, scale_dtype=torch.uint8, zp_dtype=torch.uint8)
    scheme = QuantizationScheme(targets=["Linear"], weights=args)
    sd = {"weight": w, "weight_scale": s}
    ref = before["mxfp4-pack-quantized"].compress(sd, scheme)
    got = BaseCompressor.get_value_from_registry("mxfp4-pack-quantized").compress(sd, scheme)
    assert ref.keys() == got.keys() and all(torch.equal(ref[k], got[k]) for k in ref)
    vals = torch.tensor([[0.5, -6.0, 0.0, -0.0]], dtype=torch.bfloat16)
    assert pack_fp4_to_uint8(vals).tolist() == [[0xF1, 0x80]]  # CPU tensor: the backend's req fails, upstream's body runs
    ct_amd.uninstall()
    for f in fmts:
        assert BaseCompressor.get_value_from_registry(f) is before[f]


def test_install_rebinds_the_by_name_bindings_and_uninstall_restores_them(upstream):
    """upstream's own tests import the codec classes by name (test_pack_quant.py:17-20); install() points those names inside
    upstream's modules at the HIP subclasses, uninstall() puts the originals back"""
    ct, ct_amd = upstream
    import compressed_tensors.compressors as pkg
    import compressed_tensors.compressors.pack_quantized.base as base_mod

    orig = pkg.PackedQuantizationCompressor
    assert base_mod.PackedQuantizationCompressor is orig
    ct_amd.install()
    assert pkg.PackedQuantizationCompressor is not orig and issubclass(pkg.PackedQuantizationCompressor, orig)
    assert base_mod.PackedQuantizationCompressor is pkg.PackedQuantizationCompressor
    assert pkg.IntQuantizationCompressor.__name__.endswith("MI355X")
    ct_amd.uninstall()
    assert pkg.PackedQuantizationCompressor is orig and base_mod.PackedQuantizationCompressor is orig
    ct_amd.install(rebind_names=False)
    assert pkg.PackedQuantizationCompressor is orig
    from compressed_tensors.compressors import BaseCompressor

    assert BaseCompressor.get_value_from_registry("pack-quantized") is not orig


def _w4_model(ct, n=3, rows=32, cols=256):
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme

    torch.manual_seed(0)
    args = QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = QuantizationScheme(targets=["Linear"], weights=args)  # one scheme object for the group, as apply_quantization_config attaches it
    model = torch.nn.Sequential(*[torch.nn.Linear(cols, rows, bias=False).to(torch.bfloat16) for _ in range(n)])
    for lin in model:
        w = lin.weight.data
        lin.quantization_scheme = scheme
        lin.register_parameter("weight_scale", torch.nn.Parameter((w.reshape(rows, -1, 128).abs().amax(-1).float() / 7.5).to(torch.bfloat16), requires_grad=False))
        lin.register_parameter("weight_zero_point", torch.nn.Parameter(torch.zeros(rows, cols // 128, dtype=torch.int8), requires_grad=False))
    return model


def test_install_wraps_the_model_compressor_loops_and_uninstall_restores_them(upstream):
    """install() puts batched `compress_modules` / `decompress_modules` behind upstream's ModelCompressor.compress_model /
    decompress_model (model_compressor.py:167-169,196-198); with CPU tensors every module still goes through upstream's own
    code, module by module, and the model ends exactly as upstream leaves it; uninstall() restores the original methods"""
    import copy

    ct, ct_amd = upstream
    from compressed_tensors import ModelCompressor
    from compressed_tensors.quantization import QuantizationStatus

    orig_c, orig_d = ModelCompressor.compress_model, ModelCompressor.decompress_model
    ref_model = _w4_model(ct)
    got_model = copy.deepcopy(ref_model)
    ModelCompressor().compress_model(ref_model)
    ct_amd.install()
    assert ModelCompressor.compress_model is not orig_c and ModelCompressor.decompress_model is not orig_d
    mc = ModelCompressor()
    mc.compress_model(got_model)
    assert hasattr(got_model, "ct_decompress_hook")
    for a, b in zip(ref_model, got_model):
        assert list(a._parameters) == list(b._parameters)
        for k in a._parameters:
            assert (a._parameters[k] is None and b._parameters[k] is None) or torch.equal(a._parameters[k], b._parameters[k]), k
        assert b.quantization_status == QuantizationStatus.COMPRESSED and type(b.quantization_status) is QuantizationStatus
    mc.compress_model(got_model, skip_compressed=True)  # nothing left to do; must not raise
    mc.decompress_model(got_model)
    orig_d(ModelCompressor(), ref_model)  # upstream's own loop on the model upstream compressed
    for a, b in zip(ref_model, got_model):
        assert set(a._parameters) == set(b._parameters) and torch.equal(a.weight, b.weight)
        assert b.quantization_status == QuantizationStatus.DECOMPRESSED
    assert not hasattr(got_model, "ct_decompress_hook")
    ct_amd.uninstall()
    assert ModelCompressor.compress_model is orig_c and ModelCompressor.decompress_model is orig_d


def test_install_patch_functions_rebinds_and_falls_through_on_cpu(upstream):
    """install(patch_functions=True) rebinds pack_to_int32 / unpack_from_int32 in BOTH helpers and pack_quantized.base
    (base.py:11-14 binds them by name) and dequantize / fake_quantize in lifecycle.forward; CPU tensors reach the originals;
    uninstall() restores every binding"""
    ct, ct_amd = upstream
    import compressed_tensors.compressors.pack_quantized.base as base_mod
    import compressed_tensors.compressors.pack_quantized.helpers as helpers_mod
    import compressed_tensors.quantization.lifecycle.forward as forward_mod
    from compressed_tensors.quantization import QuantizationArgs

    originals = (helpers_mod.pack_to_int32, helpers_mod.unpack_from_int32, forward_mod.dequantize, forward_mod.fake_quantize)
    ct_amd.install(patch_functions=True)
    assert helpers_mod.pack_to_int32 is base_mod.pack_to_int32 is not originals[0]
    assert helpers_mod.unpack_from_int32 is base_mod.unpack_from_int32 is not originals[1]
    assert forward_mod.dequantize is not originals[2] and forward_mod.fake_quantize is not originals[3]
    assert helpers_mod.pack_to_int32._ct_original is originals[0]
    q = torch.randint(-8, 8, (4, 64), dtype=torch.int8)
    packed = helpers_mod.pack_to_int32(q, 4)
    assert torch.equal(packed, originals[0](q, 4)) and torch.equal(helpers_mod.unpack_from_int32(packed, 4, q.shape), q)
    with pytest.raises(ValueError):
        helpers_mod.pack_to_int32(q.to(torch.int32), 4)  # upstream's own argument error (helpers.py:36-37)
    args = QuantizationArgs(num_bits=8, strategy="channel", symmetric=True)
    x = torch.randn(4, 64, dtype=torch.bfloat16)
    s = (x.abs().amax(-1, keepdim=True).float() / 127).to(torch.bfloat16)
    z = torch.zeros(4, 1, dtype=torch.int8)
    assert torch.equal(forward_mod.fake_quantize(x, s, z, args), originals[3](x, s, z, args))
    ct_amd.uninstall()
    assert (helpers_mod.pack_to_int32, helpers_mod.unpack_from_int32, forward_mod.dequantize, forward_mod.fake_quantize) == originals
    assert base_mod.pack_to_int32 is originals[0] and base_mod.unpack_from_int32 is originals[1]


def test_patch_functions_rescans_modules_imported_since_the_last_install(upstream):
    """`_patch_functions` ran once only — an upstream module imported after the first install(patch_functions=True)
    kept the eager functions.  Every install() now re-scans sys.modules; the wrappers are made once, nothing is recorded twice."""
    import types

    ct, ct_amd = upstream
    import compressed_tensors.compressors.pack_quantized.helpers as helpers_mod
    import compressed_tensors.quantization.lifecycle.forward as forward_mod

    originals = (helpers_mod.pack_to_int32, forward_mod.dequantize)
    ct_amd.install(patch_functions=True)
    wrappers = (helpers_mod.pack_to_int32, forward_mod.dequantize)
    n_first = len(ct_amd._FN_REBOUND)
    late = types.ModuleType("compressed_tensors._imported_late")
    late.pack_to_int32, late.dequantize = originals  # what `from ... import pack_to_int32` before install() leaves behind
    sys.modules[late.__name__] = late
    try:
        ct_amd.install(patch_functions=True)
        assert (late.pack_to_int32, late.dequantize) == wrappers
        assert (helpers_mod.pack_to_int32, forward_mod.dequantize) == wrappers  # not wrapped a second time
        assert len(ct_amd._FN_REBOUND) == n_first + 2
        ct_amd.install(patch_functions=True)
        assert len(ct_amd._FN_REBOUND) == n_first + 2
        ct_amd.uninstall()
        assert (late.pack_to_int32, late.dequantize) == originals and (helpers_mod.pack_to_int32, forward_mod.dequantize) == originals
        ct_amd.install(patch_functions=True)  # a fresh cycle builds fresh wrappers around the ORIGINALS, not around stale wrappers
        assert helpers_mod.pack_to_int32._ct_original is originals[0]
    finally:
        del sys.modules[late.__name__]


def test_offloaded_modules_are_left_to_upstreams_per_module_path(upstream):
    """a module whose `_parameters` is an upstream OffloadCache (offload/cache/base.py:14, a MutableMapping that
    onloads on access) must not be probed or batched by the wrapped ModelCompressor loops: `_ct_split` hands it to upstream's
    per-module path without reading a single entry"""
    from collections.abc import MutableMapping

    ct, ct_amd = upstream
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme

    class RecordingCache(MutableMapping):  # stands in for OffloadCache: every read is an onload
        def __init__(self, data):
            self.data, self.reads = dict(data), []

        def __getitem__(self, k):
            self.reads.append(k)
            return self.data[k]

        def __setitem__(self, k, v):
            self.data[k] = v

        def __delitem__(self, k):
            del self.data[k]

        def __iter__(self):
            return iter(self.data)

        def __len__(self):
            return len(self.data)

    ct_amd.install()
    cls = BaseCompressor.get_value_from_registry("pack-quantized")
    scheme = QuantizationScheme(targets=["Linear"], weights=QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group"))
    plain, off = torch.nn.Linear(256, 8, bias=False), torch.nn.Linear(256, 8, bias=False)
    for m in (plain, off):
        m.quantization_scheme = scheme
    cache = RecordingCache(off._parameters)
    off.__dict__["_parameters"] = cache
    ours, rest = cls._ct_split([plain, off], ("weight",))
    assert off in rest and off not in ours and cache.reads == []
    assert plain in rest  # a CPU module: upstream's code as well, but it WAS probed (a plain dict lookup is free)
