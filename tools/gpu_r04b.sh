#!/bin/bash
# round-4 GPU call B: bitmask kernel with progressive tile consumption (variants + stamps), marlin XCD remap, GPU tests, driver's bench command
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
: > $O/bmx.jsonl
for x in "1:0:1" "1:0:0" "0:0:1" "2:4:1" "2:8:1" "2:10:1"; do
  CT_BM_X=$x timeout 200 python tools/exp_r04.py bmx >> $O/bmx.jsonl 2>> $O/bmx.err
done
cat $O/bmx.jsonl
CT_BITMASK_RESIDENT=3 CT_BM_X=1:0:1 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_110.json 2>> $O/bmx.err
CT_BITMASK_RESIDENT=3 CT_BM_X=0:0:1 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_001.json 2>> $O/bmx.err
CT_BITMASK_RESIDENT=3 CT_BM_X=2:8:1 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_281.json 2>> $O/bmx.err
for m in 0 1; do CT_M24_X=$m timeout 300 python tools/exp_r04.py marlin; done > $O/marlin.jsonl 2> $O/marlin.err; cat $O/marlin.jsonl
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04b/bench.json"))
    print("value", r["value"], "frac", r["roofline"]["frac"], "cfg", {k:r["config"][k] for k in ("value_one_stream","ranks_seen","per_rank_GBps")})
    for k in ("bitmask","marlin24","tinyllama_checkpoint"):
        v=r.get(k,{})
        print(k, {kk:vv for kk,vv in v.items() if any(t in kk for t in ("api","_us","ms_","error"))})
    print("cpu_baseline", {k:r["cpu_baseline"].get(k) for k in ("value","cores","kind","sample")})
except Exception as e:
    print("bench parse failed", e)
PY
