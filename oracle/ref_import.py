"""Import shim for the upstream reference (TEST INFRASTRUCTURE ONLY).

The reference at /root/reference is pure Python but (a) imports `loguru`, which is
not installed in this image, and (b) expects a setuptools_scm-generated
`compressed_tensors/version.py`.  This module installs two tiny stand-ins in
`sys.modules` and puts /root/reference/src on sys.path so the reference can be
imported *in the build container* to (1) pin the oracle and (2) generate the golden
vectors under tests/golden/.  /root/reference does not exist on the GPU box; there the
archive that `oracle/stage_ref.py` packed in the build container (oracle/_ref/, git-ignored,
travels with the gpurun snapshot) is unpacked into a temp directory and imported instead, so
that the HIP path can meet the reference itself on the MI355X.  `available()` tells callers
whether either source exists; `root()` is the directory that holds `src/` and `tests/`.

Never imported by the product package (compressed_tensors_amd).
"""
import os
import sys
import types

import tempfile

LIVE_ROOT = "/root/reference"
ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference_stage.tar.gz")

# set at import time, before anything can import the reference: never drop __pycache__ into the read-only tree
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

_ROOT = None


def staged() -> bool:
    return os.path.exists(ARCHIVE)


def available() -> bool:
    return os.path.isdir(os.path.join(LIVE_ROOT, "src", "compressed_tensors")) or staged()


def root() -> str:
    """the live tree when it exists (build container), else the staged archive unpacked once per content hash under $TMPDIR"""
    global _ROOT
    if _ROOT is not None:
        return _ROOT
    if os.path.isdir(os.path.join(LIVE_ROOT, "src", "compressed_tensors")) and os.environ.get("CT_REF_FORCE_STAGED") != "1":
        _ROOT = LIVE_ROOT
        return _ROOT
    if not staged():
        raise RuntimeError("reference sources are not present on this machine (no /root/reference, no oracle/_ref archive)")
    import hashlib
    import tarfile

    with open(ARCHIVE, "rb") as f:
        tag = hashlib.sha256(f.read()).hexdigest()[:16]
    dst = os.path.join(tempfile.gettempdir(), f"ct_reference_stage_{tag}")
    if not os.path.exists(os.path.join(dst, "STAGED_FROM")):
        tmp = tempfile.mkdtemp(prefix="ct_reference_stage_", dir=tempfile.gettempdir())
        with tarfile.open(ARCHIVE, "r:gz") as tar:
            for m in tar.getmembers():  # plain relative file names only (the archive is ours, but check anyway)
                if m.name.startswith(("/", "..")) or ".." in m.name.split("/") or not (m.isfile() or m.isdir()):
                    raise RuntimeError(f"unexpected archive member {m.name!r}")
            tar.extractall(tmp)
        try:
            os.rename(tmp, dst)
        except OSError:  # another process won the race
            import shutil

            shutil.rmtree(tmp, ignore_errors=True)
    _ROOT = dst
    return _ROOT


def reference_src() -> str:
    return os.path.join(root(), "src")


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
    src = reference_src()
    if src not in sys.path:
        sys.path.insert(0, src)
    import compressed_tensors  # noqa: E402

    return compressed_tensors
