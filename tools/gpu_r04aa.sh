#!/bin/bash
# round 4, run AA: asymmetric int4 modules through the C++ host loop — smoke, full GPU suite, the bench twice (CPU baseline last)
export TMPDIR=/tmp
O=gpurun_out/r04aa; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for i in 1; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$i.json 2> $O/bench$i.err; echo "bench rc=$?"
  python - <<PY
import json
r = json.loads(open("$O/bench$i.json").read().strip().splitlines()[-1])
t = r["tinyllama_checkpoint"]; a = t["api"]
print("value", r["value"], "frac", r["roofline"]["frac"], r["roofline"].get("traffic_source"), "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["kind"])
print("api", a["ms_both"], a["api_over_kernels"], a["ms_host_until_compress_model_returns"], a["ms_host_until_decompress_model_returns"])
print("bitmask", r["bitmask"]["compress_us"], r["bitmask"]["api_compress_us"], r["bitmask"]["decompress_us"], "marlin", r["marlin24"]["kernels_us"], r["marlin24"]["compress_us_default"], r["marlin24"]["compress_us_deferred_check"])
PY
done
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04aa/bench1.json").read().strip().splitlines()[-1])
print("asym api", r["tinyllama_checkpoint"]["api"].get("asymmetric"))
PY
