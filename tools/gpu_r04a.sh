#!/bin/bash
# round-4 GPU call A (run through gpurun from the repo root): host microbench, bitmask variants + stamps, GPU tests, driver's bench command
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 300 python tools/exp_r04.py host > $O/host.json 2> $O/host.err; echo "host rc=$?"
: > $O/bmx.jsonl
for x in "1:0:0" "1:0:1" "0:0:1" "2:8:1" "2:12:1" "2:16:1" "2:12:0" "2:6:1"; do
  CT_BM_X=$x timeout 200 python tools/exp_r04.py bmx >> $O/bmx.jsonl 2>> $O/bmx.err
done
CT_BITMASK_RESIDENT=3 CT_BM_X=1:0:0 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_base.json 2>> $O/bmx.err
CT_BITMASK_RESIDENT=3 CT_BM_X=1:0:1 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_round.json 2>> $O/bmx.err
CT_BITMASK_RESIDENT=3 CT_BM_X=2:12:1 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_grad.json 2>> $O/bmx.err
cat $O/bmx.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04a/bench.json"))
    print("value", r["value"], "frac", r["roofline"]["frac"])
    for k in ("bitmask","marlin24","tinyllama_checkpoint"):
        v=r.get(k,{})
        print(k, {kk:vv for kk,vv in v.items() if any(t in kk for t in ("api","_us","ms_","error"))})
except Exception as e:
    print("bench parse failed", e)
PY
