#!/bin/bash
# round-4 GPU call D: host breakdown of the model API, marlin with the scale load first, shipped bitmask kernel (+ stamps, SQ / traffic passes), GPU tests, bench
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 300 python tools/exp_r04.py hostmodel > $O/hostmodel.json 2> $O/hostmodel.err; echo "hostmodel rc=$?"; cat $O/hostmodel.json
timeout 300 python tools/exp_r04.py marlin > $O/marlin.json 2> $O/marlin.err; cat $O/marlin.json
timeout 200 python tools/exp_r04.py bmx > $O/bmx.json 2>> $O/bmx.err; cat $O/bmx.json
CT_BITMASK_RESIDENT=3 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps.json 2>> $O/bmx.err
( MODES=1 bash tools/pmc_res.sh ) > $O/bitmask_pmc.txt 2>&1; cd $R; tail -12 $O/bitmask_pmc.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04d/bench.json"))
    print("value", r["value"], "frac", r["roofline"]["frac"])
    for k in ("bitmask","marlin24","tinyllama_checkpoint"):
        v=r.get(k,{})
        print(k, {kk:vv for kk,vv in v.items() if any(t in kk for t in ("api","_us","ms_","error"))})
except Exception as e:
    print("bench parse failed", e)
PY
cat gpurun_out/convert_rate.json 2>/dev/null
