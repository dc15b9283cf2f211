"""oracle/stage_ref.py + oracle/ref_import.py + tests/ref_suite/ct_ref_plugin.py on the CPU: the archive that carries the
reference to the GPU box unpacks and imports, and the reference's own codec tests pass under install() with CPU tensors
(everything is handed to upstream: no launch).  The GPU half is tests/test_gpu_reference_suite.py."""
import json
import os
import subprocess
import sys

import pytest

import ref_import

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="no reference on this machine")


def test_stage_recipe_is_idempotent_and_lists_only_reference_files():
    if not os.path.isdir("/root/reference/src/compressed_tensors"):
        pytest.skip("the recipe runs where /root/reference exists")
    import tarfile

    import stage_ref

    path = stage_ref.stage()
    first = os.path.getmtime(path)
    assert stage_ref.stage() == path and os.path.getmtime(path) == first  # unchanged content: not rebuilt
    names = tarfile.open(path).getnames()
    assert "src/compressed_tensors/compressors/pack_quantized/helpers.py" in names
    assert "tests/test_compressors/test_compress_decompress_module.py" in names and "STAGED_FROM" in names
    assert all(n == "STAGED_FROM" or n.startswith(("src/compressed_tensors/", "tests/")) for n in names)
    # git never sees it
    r = subprocess.run(["git", "check-ignore", "-q", "oracle/_ref/reference_stage.tar.gz"], cwd=ROOT)
    assert r.returncode == 0, "oracle/_ref must stay git-ignored"
    if os.path.exists(os.path.join(ROOT, ".gpurunignore")):
        assert "oracle/_ref" not in open(os.path.join(ROOT, ".gpurunignore")).read()


def test_staged_archive_imports_and_runs_the_references_codec_tests(tmp_path):
    if not ref_import.staged():
        pytest.skip("no staged archive (run __graft_entry__.build() in the build container)")
    report = str(tmp_path / "rep.json")
    env = dict(os.environ, CT_REF_FORCE_STAGED="1", PYTHONDONTWRITEBYTECODE="1", CT_REF_INSTALL="1", CT_REF_REPORT=report,
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "ref_suite"), ROOT]))
    root = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'oracle'); import ref_import; print(ref_import.root())"],
                          cwd=ROOT, env=env, capture_output=True, text=True, check=True).stdout.strip()
    assert ("oracle/_ref/unpacked_" in root or "ct_reference_stage_" in root) and (os.stat(root).st_mode & 0o022) == 0 and os.path.exists(os.path.join(root, "tests", "testing_utils.py"))
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "ct_ref_plugin", "-p", "no:cacheprovider", "-q",
                        "tests/test_compressors/test_pack_quant.py", "tests/test_compressors/test_int_quant.py"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    rep = json.load(open(report))
    assert r.returncode == 0 and not rep["failed"] and len(rep["passed"]) >= 100, r.stdout[-2000:]
    assert not rep["launches"]  # CPU tensors: the HIP subclass defers to upstream
    # every staged test module the GPU suite runs must at least IMPORT from the archive alone (a helper module missing from the
    # archive only shows on the GPU box otherwise: tests.test_offload.conftest did)
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "ct_ref_plugin", "-p", "no:cacheprovider", "-q", "--collect-only",
                        "tests/test_compressors/test_compress_decompress_module.py", "tests/test_compressors/test_packed_asym_decompression.py",
                        "tests/test_compressors/model_compressors/test_model_compressor.py", "tests/test_compressors/test_fp8_quant.py",
                        "tests/test_compressors/test_fp4_quant.py", "tests/test_compressors/test_mxfp4_quant.py", "tests/test_compressors/test_mxfp8_quant.py",
                        "tests/test_compressors/test_fp4_optimizations.py", "tests/test_quantization/lifecycle/test_forward.py"],
                       cwd=root, env=dict(env, CT_REF_INSTALL="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "error" not in r.stdout.lower().split("warnings")[0][-300:], r.stdout[-3000:]
