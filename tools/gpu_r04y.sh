#!/bin/bash
# round 4, run Y: the host extension built with -DNDEBUG and touching its own TLS (glibc BZ 19924): GPU suite, the bench twice, and the bad order of runs Q-W
export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
show() {
  python - <<PY
import json
r = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
t = r["tinyllama_checkpoint"]; a = t["api"]
print("$1 value", r["value"], "frac", r["roofline"]["frac"], "cpu", r.get("cpu_baseline", {}).get("value"))
print("  api", a["ms_both"], a["api_over_kernels"], a["ms_host_until_compress_model_returns"], a["ms_host_until_decompress_model_returns"],
      "bitmask", r["bitmask"]["compress_us"], r["bitmask"]["api_compress_us"], "marlin", r["marlin24"]["kernels_us"], r["marlin24"]["compress_us_default"], r["marlin24"]["compress_us_deferred_check"])
PY
}
for i in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$i.json 2> $O/bench$i.err; echo "bench rc=$?"; show bench$i
done
timeout 600 python tools/ab_bench.py cpufirst --gpus 1 --steps 20 --warmup 5 > $O/cpufirst.json 2> $O/cpufirst.err; echo "cpufirst rc=$?"; show cpufirst
