"""Import shim for the upstream reference (TEST INFRASTRUCTURE ONLY).

The reference at /root/reference is pure Python but (a) imports `loguru`, which is
not installed in this image, and (b) expects a setuptools_scm-generated
`compressed_tensors/version.py`.  This module installs two tiny stand-ins in
`sys.modules` and puts /root/reference/src on sys.path so the reference can be
imported *in the build container* to (1) pin the oracle and (2) generate the golden
vectors under tests/golden/.  /root/reference does not exist on the GPU box, so
nothing that runs there may import this module; `available()` tells callers.

Never imported by the product package (compressed_tensors_amd).
"""
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"

# set at import time, before anything can import the reference: never drop __pycache__ into the read-only tree
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "compressed_tensors"))


def _install_loguru_stub():
    if "loguru" in sys.modules:
        return
    mod = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            def _noop(*a, **k):
                return self if name in ("bind", "opt", "patch") else None

            return _noop

    mod.logger = _Logger()
    sys.modules["loguru"] = mod


def _install_version_stub():
    name = "compressed_tensors.version"
    if name in sys.modules:
        return
    mod = types.ModuleType(name)
    mod.__version__ = "0.0.0+reference"
    mod.version = mod.__version__
    mod.__all__ = ["__version__", "version"]
    sys.modules[name] = mod


def import_reference():
    """Returns the imported upstream `compressed_tensors` package."""
    if not available():
        raise RuntimeError("reference sources are not present on this machine")
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree
    _install_loguru_stub()
    _install_version_stub()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import compressed_tensors  # noqa: E402

    return compressed_tensors
