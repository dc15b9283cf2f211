#!/bin/bash
# round 4, run R: native host side of the waiting calls + streamed ModelCompressor head: smoke, A/B, full GPU suite, bench
export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 400 python tools/exp_r04.py hostab > $O/hostab.jsonl 2> $O/hostab.err; echo "hostab rc=$?"; cat $O/hostab.jsonl; tail -3 $O/hostab.err
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r=json.loads(open("gpurun_out/r04r/bench.json").read().strip().splitlines()[-1])
    print("value", r["value"], "frac", r["roofline"]["frac"])
    for k in ("bitmask","marlin24","tinyllama_checkpoint"):
        v=r.get(k,{})
        print(k, {kk:vv for kk,vv in v.items() if any(t in kk for t in ("api","_us","ms_","error"))})
except Exception as e:
    print("bench parse failed", e)
PY
