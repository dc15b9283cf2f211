#!/usr/bin/env python3
"""dev helper: VGPR / SGPR / occupancy / LDS of the kernels of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py compressed_tensors_amd/csrc/ct_quant.hip [substring ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
src = sys.argv[1]; want = sys.argv[2:]
r = subprocess.run([ge.HIPCC, *ge.HIP_FLAGS, "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
filt = "c++filt"
for b in blocks:
    name = b.split()[0]
    dem = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void ct::", "", dem).split("(")[0]
    if want and not any(w in dem for w in want):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    occ, lds = g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{dem[:70]:70s} VGPR {g('VGPRs'):>4} SGPR {g('TotalSGPRs'):>4} waves/SIMD {occ:>2} spill {g('VGPRs Spill'):>3} LDS {lds:>6}")
