#!/usr/bin/env python3
"""VGPR / SGPR / LDS / spill figures of the gfx950 kernels in one object file of the build (the code object's notes).

    python tools/kernel_resources.py ct_sparse.o [name-substring]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
KEYS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".vgpr_spill_count", ".private_segment_fixed_size")


def main():
    obj = sys.argv[1]
    if not os.path.exists(obj):
        obj = os.path.join(ROOT, "compressed_tensors_amd", "csrc", "build", obj)
    needle = sys.argv[2] if len(sys.argv) > 2 else ""
    tmp = tempfile.mkdtemp(prefix="ct_res_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f]
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, cos[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, cur = [], None
    for line in notes.splitlines():
        if re.match(r"^\s*- \.\w+:", line) and ".args" not in line and not re.match(r"^\s*- \.(address_space|offset|size|value_kind|name|actual_access|is_const)", line):
            cur = {}  # a new kernel entry of amdhsa.kernels (keys in alphabetical order: .agpr_count first)
            kernels.append(cur)
        m = re.search(r"^\s*-?\s*\.name:\s+(_Z\S+)", line)
        if m and cur is not None and ".name" not in cur:
            cur[".name"] = m.group(1)
        for k in KEYS:
            m = re.search(re.escape(k) + r":\s+(\d+)", line)
            if m and cur is not None:
                cur[k] = int(m.group(1))
    kernels = [k for k in kernels if ".name" in k]
    demangle = subprocess.run(["c++filt"], input="\n".join(k[".name"] for k in kernels), capture_output=True, text=True).stdout.splitlines()
    for k, d in zip(kernels, demangle):
        if needle in d:
            d = re.sub(r"^void ", "", d).split("(")[0]
            print(f"{d[:90]:90s} vgpr {k.get('.vgpr_count', '?'):>3} agpr {k.get('.agpr_count', 0):>3} sgpr {k.get('.sgpr_count', '?'):>3} "
                  f"lds {k.get('.group_segment_fixed_size', '?'):>6} spill {k.get('.vgpr_spill_count', 0)} scratch {k.get('.private_segment_fixed_size', 0)}")


if __name__ == "__main__":
    main()
