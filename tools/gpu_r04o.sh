#!/bin/bash
# round-4 GPU call O: tapering experiment for the resident bitmask kernel (half-size workgroups at the end of the launch)
O=gpurun_out/r04o; mkdir -p $O
export TMPDIR=/tmp
for t in 0 1 2 3 0 1; do CT_BM_TAPER=$t timeout 200 python tools/exp_r04.py bmx | sed "s/^{/{\"taper\": $t, /"; done > $O/bmx.jsonl 2> $O/bmx.err; cat $O/bmx.jsonl
CT_BITMASK_RESIDENT=3 CT_BM_TAPER=1 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_taper1.json 2>> $O/bmx.err
CT_BM_TAPER=1 CT_BITMASK_RESIDENT=1 timeout 400 python tools/exp_r02.py bmres 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print({k:v for k,v in r.items() if not k.startswith('stamps')})"
