#!/bin/bash
# round-4 GPU call K: full GPU test suite + the driver's bench command twice + host breakdown (consistent tree: host path with in-C++ scheme lookups)
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/exp_r04.py hostmodel > $O/hostmodel.json 2> $O/hostmodel.err; cat $O/hostmodel.json
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_run2.json 2> $O/bench_run2.err; echo "bench2 rc=$?"
python - <<'PY'
import json
for f in ("bench.json","bench_run2.json"):
    try:
        r=json.load(open("gpurun_out/r04k/"+f))
        print(f, "value", r["value"], "frac", r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], r["roofline"]["traffic_source"][-34:])
        print("  bitmask", {k:r["bitmask"].get(k) for k in ("compress_us","decompress_us","api_compress_us","api_decompress_us")})
        print("  marlin", {k:r["marlin24"].get(k) for k in ("kernels_us","compress_us_default","compress_us_deferred_check")})
        a=r["tinyllama_checkpoint"]["api"]; print("  api", a.get("ms_both"), a.get("api_over_kernels"), a.get("ms_host_until_compress_model_returns"), a.get("ms_host_until_decompress_model_returns"), r["tinyllama_checkpoint"]["ms_whole_checkpoint"], a.get("error"))
        print("  cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
    except Exception as e:
        print(f, "parse failed", e)
PY
