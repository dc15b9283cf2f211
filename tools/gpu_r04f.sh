#!/bin/bash
# round-4 GPU call F: activation-ordered W4 kernels with the scale entry requested first; bitmask kernel back at the run-D state
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "gidx or actorder or g_idx or golden or fuzz or compressor" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 200 python tools/exp_r04.py bmx > $O/bmx.json 2> $O/bmx.err; cat $O/bmx.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04f/bench.json"))
    print("value", r["value"], "frac", r["roofline"]["frac"])
    ko=r.get("kernels_other",{})
    for k,v in ko.items():
        if isinstance(v,dict): print(k, {kk:vv for kk,vv in v.items() if "_us" in kk or "frac" in kk or "equals" in kk})
    for k in ("bitmask","marlin24","tinyllama_checkpoint"):
        v=r.get(k,{})
        print(k, {kk:vv for kk,vv in v.items() if any(t in kk for t in ("api","_us","ms_","error"))})
except Exception as e:
    print("bench parse failed", e)
PY
