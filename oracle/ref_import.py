"""Import shim for the upstream reference (TEST INFRASTRUCTURE ONLY).

The reference at /root/reference is pure Python but (a) imports `loguru`, which is
not installed in this image, and (b) expects a setuptools_scm-generated
`compressed_tensors/version.py`.  This module installs two tiny stand-ins in
`sys.modules` and puts /root/reference/src on sys.path so the reference can be
imported *in the build container* to (1) pin the oracle and (2) generate the golden
vectors under tests/golden/.  /root/reference does not exist on the GPU box; there the
archive that `oracle/stage_ref.py` packed in the build container (oracle/_ref/, git-ignored,
travels with the gpurun snapshot) is unpacked and imported instead, so
that the HIP path can meet the reference itself on the MI355X (unpacked inside oracle/_ref/, owner-only).  `available()` tells callers
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
    # unpacked next to the archive, inside the repo's own git-ignored oracle/_ref/ (mode 0700), never under a shared, predictable
    # /tmp path that another local user could pre-create and have imported; a read-only checkout falls back to a
    # private mkdtemp directory for this process
    base = os.path.dirname(ARCHIVE)
    dst = os.path.join(base, f"unpacked_{tag}")

    def _trusted(path):
        try:
            st = os.stat(path)
        except OSError:
            return False
        return st.st_uid == os.getuid() and not (st.st_mode & 0o022) and os.path.exists(os.path.join(path, "STAGED_FROM"))

    if not _trusted(dst):
        try:
            tmp = tempfile.mkdtemp(prefix="unpacking_", dir=base)
        except OSError:
            tmp, dst = tempfile.mkdtemp(prefix="ct_reference_stage_"), None  # 0700, unpredictable, ours
        with tarfile.open(ARCHIVE, "r:gz") as tar:
            for m in tar.getmembers():  # plain relative file names only (the archive is ours, but check anyway)
                if m.name.startswith(("/", "..")) or ".." in m.name.split("/") or not (m.isfile() or m.isdir()):
                    raise RuntimeError(f"unexpected archive member {m.name!r}")
            tar.extractall(tmp)
        os.chmod(tmp, 0o700)
        if dst is None:
            dst = tmp
        else:
            try:
                os.rename(tmp, dst)
            except OSError:  # the name is taken: by another process of ours that won the race — or by something we do not trust
                if _trusted(dst):
                    import shutil

                    shutil.rmtree(tmp, ignore_errors=True)
                else:
                    dst = tmp  # keep our own private copy for this process
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
