#!/bin/bash
# round 4, run Z: the waits of the native host side (word spin / hipStreamSynchronize) against the C-ABI waits, then the final tree: GPU suite + bench
export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
timeout 300 python tools/exp_r04.py hostab > $O/hostab.jsonl 2> $O/hostab.err; echo "hostab rc=$?"; cat $O/hostab.jsonl; tail -2 $O/hostab.err
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench1.json 2> $O/bench1.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04z/bench1.json").read().strip().splitlines()[-1])
t = r["tinyllama_checkpoint"]; a = t["api"]
print("value", r["value"], "frac", r["roofline"]["frac"], r["roofline"].get("traffic_source"), "cpu", r["cpu_baseline"]["value"])
print("api", a["ms_both"], a["api_over_kernels"], "bitmask", r["bitmask"]["compress_us"], r["bitmask"]["api_compress_us"], "marlin", r["marlin24"]["kernels_us"], r["marlin24"]["compress_us_default"], r["marlin24"]["compress_us_deferred_check"])
PY
