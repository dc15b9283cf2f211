#!/usr/bin/env python3
"""Stage the upstream reference as TEST INFRASTRUCTURE for the GPU box (never shipped, never imported by the product).

    python oracle/stage_ref.py            # build container: /root/reference -> oracle/_ref/reference_stage.tar.gz

The GPU box has a GPU but no /root/reference; the build container has the reference but no GPU.  So that the HIP
path can be run against the reference ITSELF on the MI355X (VERDICT r02 "missing #1"), this
recipe packs the reference's pure-Python package and the test modules that pin this path into ONE archive under
oracle/_ref/ — git-ignored (no reference source enters the history) but not gpurun-ignored (it travels with the
snapshot, like the built .so files).  `oracle/ref_import.py` unpacks it into a temp directory on a machine without
/root/reference.  What goes in (paths relative to /root/reference):

    src/compressed_tensors/                      the package (2.3 MB of .py)
    tests/{__init__,conftest,mock_observer,testing_utils}.py
    tests/test_compressors/*.py                  incl. test_compress_decompress_module.py (@requires_gpu), test_pack_quant.py,
                                                 test_int_quant.py, test_packed_asym_decompression.py
    tests/test_quantization/lifecycle/*.py       test_forward.py's accelerator-vs-CPU comparisons (:765-1150)
    tests/test_offload/conftest.py               the `torchrun` decorator test_model_compressor.py imports

`__graft_entry__.build()` runs this whenever /root/reference is present.  Nothing here is read by compressed_tensors_amd.
"""
import hashlib
import io
import os
import sys
import tarfile

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "reference_stage.tar.gz")

WANT_DIRS = ("src/compressed_tensors", "tests/test_compressors", "tests/test_quantization/lifecycle")
WANT_FILES = ("tests/__init__.py", "tests/conftest.py", "tests/mock_observer.py", "tests/testing_utils.py",
              "tests/test_quantization/__init__.py",
              "tests/test_offload/__init__.py", "tests/test_offload/conftest.py")  # test_model_compressor.py imports its `torchrun` helper


def _members():
    out = []
    for rel in WANT_FILES:
        if os.path.exists(os.path.join(REFERENCE, rel)):
            out.append(rel)
    for d in WANT_DIRS:
        base = os.path.join(REFERENCE, d)
        for root, dirs, files in os.walk(base):
            dirs[:] = sorted(x for x in dirs if x != "__pycache__")
            for f in sorted(files):
                if f.endswith((".py", ".json", ".yaml", ".yml", ".txt")):
                    out.append(os.path.relpath(os.path.join(root, f), REFERENCE))
    return sorted(set(out))


def stage(force: bool = False) -> str:
    if not os.path.isdir(os.path.join(REFERENCE, "src", "compressed_tensors")):
        if os.path.exists(ARCHIVE):
            return ARCHIVE  # GPU box: use what travelled
        raise RuntimeError("no /root/reference here and no staged archive")
    members = _members()
    h = hashlib.sha256()
    for rel in members:
        h.update(rel.encode())
        with open(os.path.join(REFERENCE, rel), "rb") as f:
            h.update(f.read())
    digest = h.hexdigest()
    stamp = ARCHIVE + ".sha256"
    if not force and os.path.exists(ARCHIVE) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return ARCHIVE
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = ARCHIVE + ".tmp"
    with tarfile.open(tmp, "w:gz") as tar:
        for rel in members:
            with open(os.path.join(REFERENCE, rel), "rb") as f:
                data = f.read()
            info = tarfile.TarInfo(rel)
            info.size, info.mode, info.mtime = len(data), 0o644, 0
            tar.addfile(info, io.BytesIO(data))
        note = f"staged from {REFERENCE} by oracle/stage_ref.py; content sha256 {digest}\n".encode()
        info = tarfile.TarInfo("STAGED_FROM")
        info.size, info.mode, info.mtime = len(note), 0o644, 0
        tar.addfile(info, io.BytesIO(note))
    os.replace(tmp, ARCHIVE)
    with open(stamp, "w") as f:
        f.write(digest)
    return ARCHIVE


if __name__ == "__main__":
    path = stage(force="--force" in sys.argv)
    print(f"{path}: {os.path.getsize(path)} bytes, {len(_members()) if os.path.isdir(REFERENCE) else '?'} files")
