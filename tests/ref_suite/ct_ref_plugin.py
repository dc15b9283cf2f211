"""pytest plugin for running the REFERENCE'S OWN test files against the HIP path (test infrastructure).

Loaded with `-p ct_ref_plugin` by tests/test_gpu_reference_suite.py in a subprocess whose cwd is the staged (or live)
reference root, so that `tests.testing_utils`, `tests.mock_observer` and the reference's conftest resolve exactly as they do
upstream.  What it does, before any reference module is imported:

  * installs the two import shims of oracle/ref_import.py (loguru, the setuptools_scm version module) and puts the
    reference's src/ on sys.path;
  * CT_REF_INSTALL=1: `compressed_tensors_amd.install.install()` — the registry swap + ImplBackend registration of
    INTEGRATION.md §A — so that `/root/reference/src/compressed_tensors/compressors/base.py:192,218` resolves to the HIP subclasses;
  * CT_REF_PATCH_FUNCTIONS=1 (with CT_REF_INSTALL=1): `install(patch_functions=True)` — pack_to_int32 / unpack_from_int32 /
    dequantize / fake_quantize rebound to the device-dispatching wrappers too;
  * CT_REF_DEFAULT_CUDA=1: `torch.set_default_device("cuda")`, which turns the reference's CPU-tensor tests
    (test_pack_quant.py, test_int_quant.py, ...) into GPU-tensor tests without editing them;
  * counts every launch that goes through the C ABI of libct_hip.so and, at session end, writes
    {"launches": {symbol: n}, "passed": [...], "failed": [...], "skipped": [...]} to $CT_REF_REPORT.
"""
import collections
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ref_import  # noqa: E402

ref_import._install_loguru_stub()
ref_import._install_version_stub()
_src = ref_import.reference_src()
if _src not in sys.path:
    sys.path.insert(0, _src)

LAUNCHES = collections.Counter()
OUTCOMES = {"passed": [], "failed": [], "skipped": []}


def _count_launches():
    from compressed_tensors_amd import _lib

    lib = _lib.load()
    for name in _lib.EXPORTED_SYMBOLS:
        if name in ("ct_abi_version", "ct_last_error") or name.endswith(("_plan", "_workspace_bytes")):
            continue
        orig = getattr(lib, name)

        def counted(*args, _orig=orig, _name=name):
            LAUNCHES[_name] += 1
            return _orig(*args)

        setattr(lib, name, counted)


def pytest_configure(config):
    import torch

    if os.environ.get("CT_REF_INSTALL") == "1":
        import compressed_tensors_amd.install as ct_amd

        ct_amd.install(patch_functions=os.environ.get("CT_REF_PATCH_FUNCTIONS") == "1")
        if torch.cuda.is_available():
            _count_launches()
    if os.environ.get("CT_REF_DEFAULT_CUDA") == "1":
        torch.set_default_device("cuda")


def pytest_runtest_logreport(report):
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        OUTCOMES.setdefault(report.outcome, []).append(report.nodeid)


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("CT_REF_REPORT")
    if path:
        with open(path, "w") as f:
            json.dump({"launches": dict(LAUNCHES), "exitstatus": int(exitstatus), **OUTCOMES}, f)
