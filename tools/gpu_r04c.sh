#!/bin/bash
# round-4 GPU call C: bitmask variants (4-wave workgroups, gradient spreads), GPU tests, bench with the C++ host loop
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
: > $O/bmx.jsonl
for x in "2:8:1:8" "2:7:1:8" "2:9:1:8" "0:0:1:4" "2:4:1:4" "2:6:1:4" "2:8:1:4" "2:10:1:4"; do
  CT_BM_X=$x timeout 200 python tools/exp_r04.py bmx >> $O/bmx.jsonl 2>> $O/bmx.err
done
cat $O/bmx.jsonl
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04c/bench.json"))
    print("value", r["value"], "frac", r["roofline"]["frac"])
    for k in ("bitmask","marlin24","tinyllama_checkpoint"):
        v=r.get(k,{})
        print(k, {kk:vv for kk,vv in v.items() if any(t in kk for t in ("api","_us","ms_","error"))})
except Exception as e:
    print("bench parse failed", e)
PY
cat gpurun_out/refsuite_upstream_model_compressor_timing.json 2>/dev/null
