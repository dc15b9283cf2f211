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


def run_reference_tests(files, *, install, default_cuda=False, report, extra=(), timeout=1500):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", CT_REF_REPORT=report, CT_REF_INSTALL="1" if install else "0",
               CT_REF_DEFAULT_CUDA="1" if default_cuda else "0",
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
