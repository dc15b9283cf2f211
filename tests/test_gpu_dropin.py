"""The drop-in route on hardware.

The GPU box has a GPU but no upstream `compressed_tensors`, the build container has upstream but no GPU, so
`install()`'s HIP branch is driven here through the same wiring (`install_into`, `make_hip_subclass`,
`quantize_backend`) against STAND-INS that behave like upstream at the seams that matter:

* codec classes whose `compress` / `decompress` are the CPU oracle and count their calls (so a GPU tensor that
  reached them — i.e. a silent fall-through — fails the test, and a CPU tensor must reach them);
* pydantic scheme objects with enum-valued `strategy` / `type` and an assignable `format`, like upstream's
  `QuantizationArgs` / `QuantizationScheme` (quant_args.py:169-429, quant_scheme.py:26-120);
* an `ImplBackend` with upstream's dispatch rule (utils/impl_backend.py:50-123) around an eager `_quantize` body,
  called with the exact broadcast shapes upstream passes (forward_helpers.py:118-177,523-546).

A last test runs the same checks against the real upstream wherever both it and a GPU exist."""
import enum
import typing

import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def cta():
    import compressed_tensors_amd as m
    from compressed_tensors_amd import _lib

    _lib.load()
    return m


# ----------------------------------------------------------------------------- upstream-shaped scheme objects
pydantic = pytest.importorskip("pydantic")


class UpType(str, enum.Enum):
    INT = "int"
    FLOAT = "float"


class UpStrategy(str, enum.Enum):
    TENSOR = "tensor"
    CHANNEL = "channel"
    GROUP = "group"


class UpFormat(enum.Enum):
    pack_quantized = "pack-quantized"
    int_quantized = "int-quantized"
    naive_quantized = "naive-quantized"


class UpArgs(pydantic.BaseModel):
    num_bits: int = 8
    type: UpType = UpType.INT
    symmetric: bool = True
    group_size: typing.Optional[int] = None
    strategy: typing.Optional[UpStrategy] = None
    block_structure: typing.Optional[typing.List[int]] = None
    dynamic: bool = False
    actorder: typing.Optional[str] = None


class UpScheme(pydantic.BaseModel):
    targets: typing.List[str]
    weights: typing.Optional[UpArgs] = None
    input_activations: typing.Optional[UpArgs] = None
    output_activations: typing.Optional[UpArgs] = None
    format: typing.Optional[UpFormat] = None


def _kw(args):
    return dict(num_bits=args.num_bits, strategy=args.strategy.value, group_size=args.group_size, symmetric=args.symmetric)


@pytest.fixture()
def wired(cta):
    """a registry table of oracle-backed stand-in codecs + an ImplBackend, with the HIP subclasses installed"""
    from compressed_tensors_amd import install as ct_install
    from compressed_tensors_amd.utils.impl_backend import ImplBackend as _Mirror

    class Backend(_Mirror):  # private dispatch tables: nothing leaks into the package's own ImplBackend
        _backends = {}
        _fn_registry = {}

    calls = {"compress": 0, "decompress": 0, "eager_quantize": 0}

    class UpPacked(cta.BaseCompressor):
        """stands in for upstream's PackedQuantizationCompressor: CPU-only arithmetic (the oracle)"""

        @classmethod
        def compression_param_names(cls, scheme):
            return ("weight_packed", "weight_scale", "weight_shape")

        @classmethod
        def can_compress(cls, module_type, scheme):
            return True

        @classmethod
        def compress(cls, state_dict, scheme):
            calls["compress"] += 1
            assert not any(t.is_cuda for t in state_dict.values() if t is not None), "a GPU tensor fell through to the upstream codec"
            return O.pack_quantized_compress(state_dict, **_kw(scheme.weights))

        @classmethod
        def decompress(cls, state_dict, scheme):
            calls["decompress"] += 1
            assert not any(t.is_cuda for t in state_dict.values() if t is not None), "a GPU tensor fell through to the upstream codec"
            a = scheme.weights
            return O.pack_quantized_decompress(state_dict, num_bits=a.num_bits, strategy=a.strategy.value, symmetric=a.symmetric)

    class UpInt(cta.BaseCompressor):
        @classmethod
        def compress(cls, state_dict, scheme):
            calls["compress"] += 1
            raise AssertionError("a GPU tensor fell through to the upstream codec")

        decompress = compress

    table = {"pack-quantized": UpPacked, "int-quantized": UpInt, "naive-quantized": UpInt}
    saved = ct_install.install_into(table, Backend)

    @Backend.entrypoint("_quantize")
    def _quantize(x, scale, zero_point, q_min, q_max, args, dtype=None, global_scale=None):
        calls["eager_quantize"] += 1  # upstream's eager body (forward_helpers.py:535-546)
        t = x / scale
        if zero_point is not None:
            t += zero_point.to(x.dtype)
        t = torch.round(torch.clamp(t, q_min, q_max))
        return t if dtype is None else t.to(dtype)

    yield dict(table=table, saved=saved, calls=calls, quantize=_quantize, Backend=Backend, install=ct_install, UpPacked=UpPacked)
    ct_install.uninstall_from(table, saved)


def _sd(sym, rows=64, cols=512, gs=128, bits=4, seed=0):
    torch.manual_seed(seed)
    w = torch.randn(rows, cols, dtype=BF16)
    scale, zp = O.calculate_qparams_minmax(w, num_bits=bits, group_size=gs, symmetric=sym)
    return {"weight": w, "weight_scale": scale, "weight_zero_point": zp}


@pytest.mark.parametrize("sym", [True, False])
def test_hip_subclass_takes_gpu_tensors_and_defers_cpu_tensors(cta, dev, wired, sym):
    """install.py make_hip_subclass: the class a format string resolves to after install() (compressors/base.py:192,218)"""
    table, calls = wired["table"], wired["calls"]
    hip = table["pack-quantized"]
    assert hip.__name__ == "UpPackedMI355X" and issubclass(hip, wired["UpPacked"]) and hip.can_compress(torch.nn.Linear, None)
    args = UpArgs(num_bits=4, group_size=128, symmetric=sym, strategy=UpStrategy.GROUP)
    scheme = UpScheme(targets=["Linear"], weights=args)
    sd = _sd(sym)
    ref_c = O.pack_quantized_compress(sd, **_kw(args))
    got = hip.compress({k: v.to(dev) for k, v in sd.items()}, scheme)
    assert calls["compress"] == 0, "GPU tensors must take the HIP branch"
    assert sorted(got) == sorted(ref_c)
    for k in ref_c:
        assert got[k].device.type == ("cpu" if k == "weight_shape" else "cuda"), k
        assert torch.equal(got[k].cpu().contiguous(), ref_c[k].contiguous()), k
    back = hip.decompress(got, scheme)
    ref_d = O.pack_quantized_decompress(ref_c, num_bits=4, strategy="group", symmetric=sym)
    assert calls["decompress"] == 0 and sorted(back) == sorted(ref_d)
    assert torch.equal(back["weight"].cpu().view(torch.int16), ref_d["weight"].view(torch.int16))
    if not sym:
        assert torch.equal(back["weight_zero_point"].cpu(), ref_d["weight_zero_point"])
    # CPU tensors: upstream's own implementation runs, exactly once each way
    got_cpu = hip.compress(sd, scheme)
    assert calls["compress"] == 1 and torch.equal(got_cpu["weight_packed"], ref_c["weight_packed"])
    hip.decompress(got_cpu, scheme)
    assert calls["decompress"] == 1


def test_compress_module_through_the_swapped_registry(cta, dev, wired):
    """upstream compress_module / decompress_module (compressors/base.py:170-219): set scheme.format (an enum assigned onto a
    pydantic scheme), look the codec up by format string, call its inherited *_module glue — with CUDA parameters"""
    table, calls = wired["table"], wired["calls"]
    from compressed_tensors_amd.quantization.quant_args import QuantizationStatus

    def compress_module(module, fmt=None):  # restated from upstream: the lookup is by string at call time
        scheme = module.quantization_scheme
        scheme.format = UpFormat(fmt or scheme.format or "pack-quantized")
        table[scheme.format.value].compress_module(module)

    def decompress_module(module):
        table[module.quantization_scheme.format.value].decompress_module(module)

    sd = _sd(False, rows=128, cols=256, seed=5)
    lin = torch.nn.Linear(256, 128, bias=True).to(BF16)
    lin.weight.data.copy_(sd["weight"])
    lin.register_parameter("weight_scale", torch.nn.Parameter(sd["weight_scale"], requires_grad=False))
    lin.register_parameter("weight_zero_point", torch.nn.Parameter(sd["weight_zero_point"], requires_grad=False))
    lin = lin.to(dev)
    bias = lin.bias
    lin.quantization_scheme = UpScheme(targets=["Linear"], weights=UpArgs(num_bits=4, group_size=128, symmetric=False, strategy=UpStrategy.GROUP))
    compress_module(lin)
    ref_c = O.pack_quantized_compress(sd, num_bits=4, strategy="group", group_size=128, symmetric=False)
    assert calls["compress"] == 0 and lin.quantization_status == QuantizationStatus.COMPRESSED
    # untouched tensors keep their storage (a fresh non-trainable Parameter around the same data: `.data` makes a new tensor
    # object on every call, so upstream's identity shortcut at utils/module.py:56-59 never fires — mirrored as is)
    assert not hasattr(lin, "weight") and lin.bias.data_ptr() == bias.data_ptr() and torch.equal(lin.bias.data, bias.data)
    assert torch.equal(lin.weight_packed.data.cpu(), ref_c["weight_packed"]) and lin.weight_packed.is_cuda
    assert torch.equal(lin.weight_zero_point.data.cpu(), ref_c["weight_zero_point"])
    assert lin.weight_shape.tolist() == [128, 256]
    decompress_module(lin)
    ref_d = O.pack_quantized_decompress(ref_c, num_bits=4, strategy="group", symmetric=False)
    assert calls["decompress"] == 0 and lin.quantization_status == QuantizationStatus.DECOMPRESSED
    assert torch.equal(lin.weight.data.cpu().view(torch.int16), ref_d["weight"].view(torch.int16)) and lin.weight.is_cuda


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("bits,sym", [(4, True), (4, False), (8, True), (3, False)])
def test_quantize_backend_with_upstreams_broadcast_shapes(cta, dev, wired, dtype, bits, sym):
    """the `_quantize` entrypoint as upstream calls it: group -> x (R, G, gs) / scale (R, G, 1) (forward_helpers.py:154-167),
    channel -> x (R, C) / scale (R, 1), tensor -> x (R, C) / scale (1,); q_min / q_max are 0-dim fp32 device tensors
    (utils/helpers.py:208-211); dtype None keeps x's float dtype (fake-quantize callers), int8 for compressors"""
    q, calls = wired["quantize"], wired["calls"]
    torch.manual_seed(bits)
    R, C, gs = 48, 384, 128
    w = torch.randn(R, C).to(dtype)
    lo, hi = torch.tensor(-(2 ** bits) / 2, device=dev), torch.tensor(2 ** bits / 2 - 1, device=dev)
    for strategy in ("group", "channel", "tensor"):
        g = gs if strategy == "group" else None
        src = w.reshape(1, -1) if strategy == "tensor" else w
        scale, zp = O.calculate_qparams_minmax(src, num_bits=bits, group_size=g, symmetric=sym)
        if strategy == "group":
            xs, ss, zs = w.reshape(R, C // gs, gs), scale.unsqueeze(-1), zp.unsqueeze(-1)
        elif strategy == "channel":
            xs, ss, zs = w, scale, zp
        else:
            scale, zp = scale.reshape(1), zp.reshape(1)
            xs, ss, zs = w, scale, zp
        args = UpArgs(num_bits=bits, symmetric=sym, strategy=UpStrategy(strategy), group_size=g)
        for out_dtype in (torch.int8, None):
            before = calls["eager_quantize"]
            got = q(xs.to(dev), ss.to(dev), zs.to(dev), lo, hi, args, dtype=out_dtype)
            assert calls["eager_quantize"] == before, "the HIP backend must accept upstream's shapes"
            ref = O.quantize(w, scale, zp, num_bits=bits, strategy=strategy, group_size=g, dtype=out_dtype)
            assert got.shape == xs.shape and got.dtype == (out_dtype or dtype) and got.is_cuda
            assert torch.equal(got.cpu().reshape(R, C).float(), ref.float()), (strategy, out_dtype)
        # zero_point=None (symmetric callers that drop it) and CPU inputs (the req fails: upstream's body runs)
        if sym:
            got = q(xs.to(dev), ss.to(dev), None, lo, hi, args, dtype=torch.int8)
            assert torch.equal(got.cpu().reshape(R, C), O.quantize(w, scale, None, num_bits=bits, strategy=strategy, group_size=g, dtype=torch.int8))
    before = calls["eager_quantize"]
    q(xs, ss, zs, lo.cpu(), hi.cpu(), args, dtype=torch.int8)
    assert calls["eager_quantize"] == before + 1
    # a layout the backend does not recognise (scale broadcast over rows) falls through as well
    q(w.to(dev), torch.ones(1, C, dtype=dtype, device=dev), None, lo, hi, args, dtype=torch.int8)
    assert calls["eager_quantize"] == before + 2
    assert "_quantize_mi355x" in wired["Backend"]._fn_registry


def test_install_is_idempotent_and_reversible(cta, wired):
    table, saved, inst = wired["table"], wired["saved"], wired["install"]
    first = dict(table)
    inst.install_into(table, wired["Backend"], saved)  # second install: same originals, fresh subclasses, no duplicate backends
    assert all(issubclass(table[f], saved[f]) and table[f] is not saved[f] for f in saved)
    assert [fn.__name__ for fn, _, _ in wired["Backend"]._backends["_quantize"]] == ["_quantize_mi355x"]
    originals = dict(saved)
    inst.uninstall_from(table, saved)
    assert all(table[f] is originals[f] for f in originals) and not saved
    inst.install_into(table, wired["Backend"], saved)  # leave it installed for the fixture's teardown
    assert set(first) == set(table)


def test_against_the_real_upstream_when_present(cta, dev):
    """runs wherever upstream `compressed_tensors` AND a GPU exist (neither this box nor the build container has both)"""
    try:
        import ref_import

        if ref_import.available():
            ref_import.import_reference()
        import compressed_tensors  # noqa: F401
    except Exception:
        pytest.skip("upstream compressed_tensors is not importable on this machine")
    from compressed_tensors.compressors import BaseCompressor, compress_module, decompress_module
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme
    from compressed_tensors.quantization.lifecycle.forward import fake_quantize, quantize

    import compressed_tensors_amd.install as ct_amd

    ct_amd.install()
    try:
        sd = _sd(False, rows=128, cols=256, seed=9)
        args = QuantizationArgs(num_bits=4, group_size=128, symmetric=False, strategy="group")
        lin = torch.nn.Linear(256, 128, bias=False).to(BF16)
        lin.weight.data.copy_(sd["weight"])
        lin.register_parameter("weight_scale", torch.nn.Parameter(sd["weight_scale"], requires_grad=False))
        lin.register_parameter("weight_zero_point", torch.nn.Parameter(sd["weight_zero_point"], requires_grad=False))
        lin = lin.to(dev)
        lin.quantization_scheme = QuantizationScheme(targets=["Linear"], weights=args)
        assert BaseCompressor.get_value_from_registry("pack-quantized").__name__.endswith("MI355X")
        compress_module(lin)
        ref_c = O.pack_quantized_compress(sd, num_bits=4, strategy="group", group_size=128, symmetric=False)
        assert torch.equal(lin.weight_packed.data.cpu(), ref_c["weight_packed"])
        decompress_module(lin)
        fq = fake_quantize(sd["weight"], sd["weight_scale"], sd["weight_zero_point"], args)
        assert torch.equal(lin.weight.data.cpu(), fq)
        got = quantize(sd["weight"].to(dev), sd["weight_scale"].to(dev), sd["weight_zero_point"].to(dev), args, dtype=torch.int8)
        assert torch.equal(got.cpu(), quantize(sd["weight"], sd["weight_scale"], sd["weight_zero_point"], args, dtype=torch.int8))
    finally:
        ct_amd.uninstall()
