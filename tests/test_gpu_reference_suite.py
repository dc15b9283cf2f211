"""The HIP path against the REFERENCE ITSELF on the MI355X (VERDICT r02 missing #1 / next #2).

`oracle/stage_ref.py` packs the reference's package and the test modules that pin this path into the git-ignored
`oracle/_ref/reference_stage.tar.gz`, which travels to the GPU box with the snapshot; `oracle/ref_import.py` unpacks and
imports it there.  Two kinds of test:

* the reference's OWN test files, unmodified, run in a subprocess (cwd = the reference root, plug-in
  tests/ref_suite/ct_ref_plugin.py) with `compressed_tensors_amd.install.install()` active, so that
  `compressors/base.py:192,218` resolves to the HIP subclasses and `_quantize` dispatches to the HIP ImplBackend backend.
  Each suite also runs WITHOUT install() as the baseline of what upstream itself does on this box: a test may fail with
  the HIP path only if upstream's own path fails it too.  The plug-in counts the launches through the C ABI — a suite that
  passes without ever reaching libct_hip.so fails here;
* install()ed GPU outputs against the reference's CPU outputs, bit for bit, on BASELINE configs 1 and 2 at full size and
  on the asymmetric / activation-ordered / channel-wise / 8-bit / 3-D variants.
"""
import json
import os
import subprocess
import sys
import zlib

import pytest
import torch

import ref_import

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN_DIR = os.path.join(ROOT, "tests", "ref_suite")

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_import.available(), reason="no reference on this machine: run oracle/stage_ref.py in the build container")]


def run_reference_tests(files, *, install, default_cuda=False, report, extra=(), timeout=1500, patch_functions=False):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", CT_REF_REPORT=report, CT_REF_INSTALL="1" if install else "0",
               CT_REF_DEFAULT_CUDA="1" if default_cuda else "0", CT_REF_PATCH_FUNCTIONS="1" if patch_functions else "0",
               PYTHONPATH=os.pathsep.join([PLUGIN_DIR, ROOT, os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "pytest", "-p", "ct_ref_plugin", "-p", "no:cacheprovider", "-q", "--no-header", "-rN", *extra, *files]
    r = subprocess.run(cmd, cwd=ref_import.root(), env=env, capture_output=True, text=True, timeout=timeout)
    try:
        rep = json.load(open(report))
    except Exception:
        raise AssertionError(f"the reference test run produced no report (rc {r.returncode}):\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
    rep["stdout_tail"] = r.stdout[-4000:]
    return rep


def _keep(rep, name):
    """leave the outcome lists where the builder / judge can read them after a gpurun call"""
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(rep, open(os.path.join(out, f"refsuite_{name}.json"), "w"), indent=1)
    except OSError:
        pass


def _check(with_hip, baseline, name, need_launches, min_passed):
    _keep(with_hip, name + "_hip")
    _keep(baseline, name + "_upstream")
    new_failures = sorted(set(with_hip["failed"]) - set(baseline["failed"]))
    assert not new_failures, f"fail with the HIP path but pass with upstream's own: {new_failures}\n{with_hip['stdout_tail']}"
    lost = sorted(set(baseline["passed"]) - set(with_hip["passed"]))
    assert not lost, f"pass upstream, did not pass with the HIP path: {lost}"
    assert len(with_hip["passed"]) >= min_passed, (len(with_hip["passed"]), with_hip["stdout_tail"])
    for sym in need_launches:
        assert with_hip["launches"].get(sym, 0) > 0, f"{sym} was never launched: the HIP branch was not taken ({with_hip['launches']})"
    assert not baseline["launches"], "the baseline run must not touch libct_hip.so"


def test_reference_module_tests_through_the_swapped_registry(tmp_path):
    """tests/test_compressors/test_compress_decompress_module.py (hard-codes "cuda" at :34): compress_module / decompress_module of
    every preset scheme incl. W4A16 with activation ordering, W4A16_ASYM, W8A16, W8A8, FP8, NVFP4, MXFP4, Linear and Embedding"""
    files = ["tests/test_compressors/test_compress_decompress_module.py"]
    hip = run_reference_tests(files, install=True, report=str(tmp_path / "hip.json"))
    up = run_reference_tests(files, install=False, report=str(tmp_path / "up.json"))
    _check(hip, up, "module", ("ct_quant_pack", "ct_unpack_dequant", "ct_quantize", "ct_dequantize"), min_passed=40)


def test_reference_codec_tests_on_gpu_tensors(tmp_path):
    """test_pack_quant.py (:160-183 round trip == fake_quantize, :186-235 packed zero points, :238-277 actorder, :346-367 3-D),
    test_int_quant.py (:44-102) and test_packed_asym_decompression.py build CPU tensors; with the default device set to "cuda"
    the same unmodified tests feed GPU tensors through the swapped registry"""
    files = ["tests/test_compressors/test_pack_quant.py", "tests/test_compressors/test_int_quant.py",
             "tests/test_compressors/test_packed_asym_decompression.py"]
    hip = run_reference_tests(files, install=True, default_cuda=True, report=str(tmp_path / "hip.json"))
    up = run_reference_tests(files, install=False, default_cuda=True, report=str(tmp_path / "up.json"))
    _check(hip, up, "codecs_cuda", ("ct_quant_pack", "ct_unpack_dequant", "ct_quantize", "ct_dequantize"), min_passed=60)
    # and with CPU tensors the subclass must hand everything to upstream: identical outcomes, no launch
    cpu = run_reference_tests(files, install=True, report=str(tmp_path / "cpu.json"))
    assert not cpu["failed"] and len(cpu["passed"]) >= 140 and not cpu["launches"], (cpu["failed"], cpu["launches"])


def test_reference_forward_tests_accelerator_vs_cpu(tmp_path):
    """tests/test_quantization/lifecycle/test_forward.py: the @requires_gpu accelerator-vs-CPU comparisons of `_quantize`
    (:617-739 fused vs sequential, :765-1150 CUDA vs CPU incl. non-contiguous inputs) now dispatch to the HIP backend"""
    files = ["tests/test_quantization/lifecycle/test_forward.py"]
    hip = run_reference_tests(files, install=True, report=str(tmp_path / "hip.json"))
    up = run_reference_tests(files, install=False, report=str(tmp_path / "up.json"))
    _check(hip, up, "forward", ("ct_quantize", "ct_quantize_fp8", "ct_quantize_fp4"), min_passed=80)


def test_reference_model_compressor_and_float_format_tests(tmp_path):
    """tests/test_compressors/model_compressors/test_model_compressor.py (ModelCompressor.compress_model / decompress_model over whole
    models, format inference, the 2-GPU cases skip themselves) and the FP8 / FP4 / MX codec tests, on GPU tensors (default device
    "cuda") through the swapped registry and the rebound class names"""
    files = ["tests/test_compressors/model_compressors/test_model_compressor.py", "tests/test_compressors/test_fp8_quant.py",
             "tests/test_compressors/test_fp4_quant.py", "tests/test_compressors/test_mxfp4_quant.py", "tests/test_compressors/test_mxfp8_quant.py",
             "tests/test_compressors/test_fp4_optimizations.py"]
    hip = run_reference_tests(files, install=True, default_cuda=True, report=str(tmp_path / "hip.json"))
    up = run_reference_tests(files, install=False, default_cuda=True, report=str(tmp_path / "up.json"))
    _check(hip, up, "model_and_float_cuda", (), min_passed=30)
    assert sum(hip["launches"].values()) > 0, "no launch reached libct_hip.so"
    plain = run_reference_tests(files, install=True, report=str(tmp_path / "plain.json"))
    plain_up = run_reference_tests(files, install=False, report=str(tmp_path / "plain_up.json"))
    _check(plain, plain_up, "model_and_float", (), min_passed=30)


def test_reference_codec_tests_with_the_plain_functions_patched(tmp_path):
    """VERDICT r03 missing #5: `install(patch_functions=True)` rebinds pack_to_int32 / unpack_from_int32 (helpers.py:20-24,104-109 and
    their by-name bindings at pack_quantized/base.py:11-14) and dequantize / fake_quantize (lifecycle/forward.py:76-181); the reference's
    own pack-quant / int-quant tests on GPU tensors then reach the pack / unpack / fake-quantize kernels directly, with the same
    outcomes as upstream's eager ops"""
    files = ["tests/test_compressors/test_pack_quant.py", "tests/test_compressors/test_int_quant.py",
             "tests/test_compressors/test_packed_asym_decompression.py"]
    hip = run_reference_tests(files, install=True, default_cuda=True, patch_functions=True, report=str(tmp_path / "hip.json"))
    up = run_reference_tests(files, install=False, default_cuda=True, report=str(tmp_path / "up.json"))
    _check(hip, up, "codecs_cuda_patched", ("ct_pack_int32", "ct_unpack_int32", "ct_fake_quantize"), min_passed=60)
    fwd = ["tests/test_quantization/lifecycle/test_forward.py"]
    hip = run_reference_tests(fwd, install=True, patch_functions=True, report=str(tmp_path / "hipf.json"))
    up = run_reference_tests(fwd, install=False, report=str(tmp_path / "upf.json"))
    _check(hip, up, "forward_patched", ("ct_quantize",), min_passed=80)


# ----------------------------------------------------------------------------- install()ed GPU outputs == the reference's CPU outputs
@pytest.fixture(scope="module")
def upstream():
    ct = ref_import.import_reference()
    import compressed_tensors_amd.install as ct_amd
    from compressed_tensors.compressors import BaseCompressor

    originals = {f: BaseCompressor.get_value_from_registry(f) for f in ("pack-quantized", "int-quantized", "naive-quantized")}
    ct_amd.install()
    yield ct, originals
    ct_amd.uninstall()


def _ref_qparams(w, args):
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


CASES = {
    # BASELINE config 1: int8 per-tensor symmetric IntQuantizationCompressor round trip, 4096x4096 bf16
    "config1_int8_tensor_4096": dict(fmt="int-quantized", shape=(4096, 4096), dtype=torch.bfloat16, args=dict(num_bits=8, strategy="tensor", symmetric=True), act=True),
    # BASELINE config 2: W4A16 pack-quantized g128, 8192x8192 bf16
    "config2_w4_g128_8192": dict(fmt="pack-quantized", shape=(8192, 8192), dtype=torch.bfloat16, args=dict(num_bits=4, strategy="group", group_size=128, symmetric=True)),
    "w4_g128_asym": dict(fmt="pack-quantized", shape=(2048, 4096), dtype=torch.bfloat16, args=dict(num_bits=4, strategy="group", group_size=128, symmetric=False)),
    "w4_g128_actorder": dict(fmt="pack-quantized", shape=(1024, 4096), dtype=torch.bfloat16, args=dict(num_bits=4, strategy="group", group_size=128, symmetric=True, actorder="group"), g_idx=True),
    "w4_g128_asym_actorder_fp16": dict(fmt="pack-quantized", shape=(512, 2048), dtype=torch.float16, args=dict(num_bits=4, strategy="group", group_size=128, symmetric=False, actorder="group"), g_idx=True),
    "w4_channel_fp16": dict(fmt="pack-quantized", shape=(1024, 4096), dtype=torch.float16, args=dict(num_bits=4, strategy="channel", symmetric=True)),
    "w8_g128": dict(fmt="pack-quantized", shape=(1024, 2048), dtype=torch.bfloat16, args=dict(num_bits=8, strategy="group", group_size=128, symmetric=True)),
    "w3_g64_asym": dict(fmt="pack-quantized", shape=(256, 1024), dtype=torch.bfloat16, args=dict(num_bits=3, strategy="group", group_size=64, symmetric=False)),
    "w4_g128_experts_3d": dict(fmt="pack-quantized", shape=(4, 256, 512), dtype=torch.bfloat16, args=dict(num_bits=4, strategy="group", group_size=128, symmetric=True),
                               compress_only=True),  # upstream's decompress cannot infer a strategy from a 3-D scale (forward.py:99-130)
    "int8_channel_asym_naive": dict(fmt="naive-quantized", shape=(1024, 2048), dtype=torch.bfloat16, args=dict(num_bits=8, strategy="channel", symmetric=False)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_installed_gpu_outputs_equal_the_references_cpu_outputs(upstream, name):
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme

    _, originals = upstream
    case = CASES[name]
    dev = torch.device("cuda:0")
    torch.manual_seed(zlib.crc32(name.encode()) % 1000)
    args = QuantizationArgs(**case["args"])
    scheme = QuantizationScheme(targets=["Linear"], weights=args, input_activations=QuantizationArgs(num_bits=8) if case.get("act") else None)
    w = torch.randn(*case["shape"], dtype=torch.float32).mul_(0.05).to(case["dtype"])
    scale, zp = _ref_qparams(w, args)
    sd = {"weight": w, "weight_scale": scale.to(case["dtype"]), "weight_zero_point": zp}
    if case.get("g_idx"):
        cols = case["shape"][-1]
        sd["weight_g_idx"] = (torch.randperm(cols) // args.group_size).to(torch.int32)
    ref_cls = originals[case["fmt"]]
    hip_cls = BaseCompressor.get_value_from_registry(case["fmt"])
    assert hip_cls is not ref_cls and issubclass(hip_cls, ref_cls) and hip_cls.__name__.endswith("MI355X")

    ref_c = ref_cls.compress(dict(sd), scheme)  # upstream's own code on CPU tensors
    hip_c = hip_cls.compress({k: v.to(dev) for k, v in sd.items()}, scheme)
    torch.cuda.synchronize()
    assert set(ref_c) == set(hip_c), (sorted(ref_c), sorted(hip_c))
    for k, v in ref_c.items():
        g = hip_c[k]
        assert g.shape == v.shape and g.dtype == v.dtype, (k, g.shape, v.shape, g.dtype, v.dtype)
        assert torch.equal(g.cpu().contiguous().view(torch.uint8), v.contiguous().view(torch.uint8)), f"{name}: compressed[{k}] differs from the reference"
        if k != "weight_shape":
            assert g.is_cuda, f"{k} left the GPU"
    if case.get("compress_only"):
        return
    ref_d = ref_cls.decompress(dict(ref_c), scheme)
    hip_d = hip_cls.decompress(dict(hip_c), scheme)
    torch.cuda.synchronize()
    assert set(ref_d) == set(hip_d), (sorted(ref_d), sorted(hip_d))
    for k, v in ref_d.items():
        g = hip_d[k].cpu().contiguous()
        assert g.shape == v.shape and g.dtype == v.dtype, (k, g.shape, v.shape, g.dtype, v.dtype)
        assert torch.equal(g.view(torch.uint8), v.contiguous().view(torch.uint8)), f"{name}: decompressed[{k}] differs from the reference"


def test_upstream_model_compressor_takes_the_batched_launches(upstream):
    """VERDICT r03 missing #4 / next #5: under install() the reference's own ModelCompressor.compress_model / decompress_model
    (model_compressor.py:138-207) hand a W4A16 model's modules to the batched entries — `ct_quant_pack_batch` /
    `ct_unpack_dequant_batch` are launched, the per-module `ct_quant_pack` / `ct_unpack_dequant` are not — and the model ends
    bit-identical to what upstream's own loop makes of the same model on the CPU"""
    import collections
    import copy

    from compressed_tensors import ModelCompressor
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme, QuantizationStatus
    from compressed_tensors_amd import _lib
    import compressed_tensors_amd.install as ct_amd

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    args = QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = QuantizationScheme(targets=["Linear"], weights=args)
    shapes = [(256, 512), (64, 512), (512, 256), (256, 512), (128, 1024)]
    cpu_model = torch.nn.Sequential(*[torch.nn.Linear(c, r, bias=False).to(torch.bfloat16) for r, c in shapes])
    for lin in cpu_model:
        w = lin.weight.data
        scale, zp = _ref_qparams(w, args)
        lin.quantization_scheme = scheme
        lin.register_parameter("weight_scale", torch.nn.Parameter(scale.to(torch.bfloat16), requires_grad=False))
        lin.register_parameter("weight_zero_point", torch.nn.Parameter(zp, requires_grad=False))
    gpu_model = copy.deepcopy(cpu_model).to(dev)
    for a, b in zip(cpu_model, gpu_model):
        b.quantization_scheme = a.quantization_scheme
    # upstream's own loop, CPU tensors (the wrapper hands CPU modules to upstream's code one by one)
    ModelCompressor().compress_model(cpu_model)

    lib = _lib.load()
    counts = collections.Counter()
    names = ("ct_quant_pack", "ct_unpack_dequant", "ct_quant_pack_batch", "ct_unpack_dequant_batch")
    saved = {n: getattr(lib, n) for n in names}
    try:
        for n in names:
            def counted(*a, _o=saved[n], _n=n):
                counts[_n] += 1
                return _o(*a)
            setattr(lib, n, counted)
        mc = ModelCompressor()
        mc.compress_model(gpu_model)
        torch.cuda.synchronize()
        assert counts["ct_quant_pack_batch"] == 1 and counts["ct_quant_pack"] == 0, dict(counts)
        for a, b in zip(cpu_model, gpu_model):
            assert set(a._parameters) == set(b._parameters)
            for k, v in a._parameters.items():
                assert (v is None and b._parameters[k] is None) or torch.equal(b._parameters[k].cpu(), v), k
            assert b.quantization_status == QuantizationStatus.COMPRESSED and type(b.quantization_status) is QuantizationStatus
            assert b.weight_packed.is_cuda and not b.weight_shape.is_cuda
        mc.decompress_model(gpu_model)
        torch.cuda.synchronize()
        assert counts["ct_unpack_dequant_batch"] == 1 and counts["ct_unpack_dequant"] == 0, dict(counts)
    finally:
        for n in names:
            setattr(lib, n, saved[n])
    ct_amd._MC_SAVED["decompress_model"](ModelCompressor(), cpu_model)  # upstream's own decompress loop
    for a, b in zip(cpu_model, gpu_model):
        assert torch.equal(b.weight.cpu().view(torch.int16), a.weight.view(torch.int16))
        assert b.quantization_status == QuantizationStatus.DECOMPRESSED


def test_upstream_model_compressor_on_a_tinyllama_tree_batched_vs_loop(upstream):
    """the same measurement bench.py makes for our ModelCompressor (`tinyllama_checkpoint.api`), for the REFERENCE's ModelCompressor under
    install(): a 154-module TinyLlama-shaped tree, compress_model + decompress_model wall time with the wrapper (batched launches) and with
    install(wrap_model_compressor=False) (upstream's loop: one launch and one full state-dict replacement per module).  Asserts the
    batched form is not slower and leaves the figures in gpurun_out/upstream_model_compressor_timing.json"""
    import time

    from compressed_tensors import ModelCompressor
    from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme
    import compressed_tensors_amd as cta
    import compressed_tensors_amd.install as ct_amd

    dev = torch.device("cuda:0")
    layer = (("q_proj", 2048, 2048), ("k_proj", 256, 2048), ("v_proj", 256, 2048), ("o_proj", 2048, 2048), ("gate_proj", 5632, 2048), ("up_proj", 5632, 2048),
             ("down_proj", 2048, 5632))
    scheme = QuantizationScheme(targets=["Linear"], weights=QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group"))
    g = torch.Generator(device=dev).manual_seed(5)
    root = torch.nn.Module()
    root.layers = torch.nn.ModuleList()
    first = None
    for _ in range(22):
        blk = torch.nn.Module()
        for name, r, c in layer:
            lin = torch.nn.Linear(c, r, bias=False, device="meta")
            w = torch.randn(r, c, dtype=torch.bfloat16, device=dev, generator=g)
            sc, zp = cta.codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
            lin.weight = torch.nn.Parameter(w, requires_grad=False)
            lin.weight_scale = torch.nn.Parameter(sc, requires_grad=False)
            lin.weight_zero_point = torch.nn.Parameter(zp, requires_grad=False)
            lin.quantization_scheme = scheme
            setattr(blk, name, lin)
            first = first or (lin, cta.codec.fake_quantize_tensor(w, sc, zp, num_bits=4, strategy="group", group_size=128))
        root.layers.append(blk)

    def timed(wrap):
        ct_amd.uninstall()
        ct_amd.install(wrap_model_compressor=wrap)
        mc = ModelCompressor()
        ts = []
        for k in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mc.compress_model(root)
            mc.decompress_model(root)
            torch.cuda.synchronize()
            if k >= 2:
                ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    try:
        ms_loop = timed(False)
        ms_batched = timed(True)
    finally:
        ct_amd.uninstall()
        ct_amd.install()  # what the module-scoped fixture expects to find
    assert torch.equal(first[0].weight.data, first[1]), "round trip through upstream's ModelCompressor != fake_quantize"
    rep = {"ms_both_batched": round(ms_batched, 3), "ms_both_upstream_loop": round(ms_loop, 3), "modules": 154,
           "what": "compressed_tensors.ModelCompressor().compress_model + .decompress_model on a TinyLlama-shaped tree under compressed_tensors_amd.install.install()"}
    print(rep)
    _keep(rep, "upstream_model_compressor_timing")
    assert ms_batched <= ms_loop * 1.05, rep
