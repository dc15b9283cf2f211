#!/bin/bash
# round-4 GPU call H: observer kernel check, full GPU test suite, the driver's bench command (twice), profile passes at the final library hash
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 300 python tools/bench_leg.py qparams_leg > $O/qparams.json 2> $O/qparams.err; cat $O/qparams.json
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_run2.json 2> $O/bench_run2.err; echo "bench2 rc=$?"
bash tools/gpu_r04g.sh > $O/profile.log 2>&1; cd $R; tail -3 $O/profile.log | cut -c1-150
python - <<'PY'
import json
for f in ("bench.json","bench_run2.json"):
    try:
        r=json.load(open("gpurun_out/r04h/"+f))
        print(f, "value", r["value"], "frac", r["roofline"]["frac"], "traffic_source", r["roofline"]["traffic_source"][-40:])
        print("  bitmask", {k:r["bitmask"].get(k) for k in ("compress_us","decompress_us","api_compress_us","api_decompress_us")})
        print("  marlin", {k:r["marlin24"].get(k) for k in ("kernels_us","compress_us_default","compress_us_deferred_check")})
        print("  qparams", r["minmax_qparams"]["us"], r["minmax_qparams"]["fused_with_compress"]["us"])
        a=r["tinyllama_checkpoint"]["api"]; print("  api", a["ms_both"], a["api_over_kernels"], r["tinyllama_checkpoint"]["ms_whole_checkpoint"])
        print("  cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
    except Exception as e:
        print(f, "parse failed", e)
PY
