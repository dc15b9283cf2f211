#!/bin/bash
# round-4 GPU call L: randomised differential run against the oracle + the resident bitmask kernel over shapes / densities, at the final library
O=gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
{ echo "# tools/fuzz_parity.py at library $(cat compressed_tensors_amd/libct_hip.so.srchash)"; timeout 200 python tools/fuzz_parity.py 80 1104 2>&1 | grep fuzz; timeout 200 python tools/fuzz_parity.py 80 2204 2>&1 | grep fuzz; } > $O/fuzz.txt; cat $O/fuzz.txt
CT_BITMASK_RESIDENT=1 timeout 400 python tools/exp_r02.py bmres > $O/bmres.json 2> $O/bmres.err; python -c "
import json; r=json.load(open('gpurun_out/r04l/bmres.json')); print({k:v for k,v in r.items() if not k.startswith('stamps')})"
