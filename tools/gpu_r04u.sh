#!/bin/bash
# round 4, run U: what moves the ModelCompressor figure of the bench (old tree 1.09 ms, current tree 1.19 ms on one lease, run T)?
export TMPDIR=/tmp
O=gpurun_out/r04u; mkdir -p $O
for m in asis pywait nogc nocpu asis; do
  timeout 400 python tools/ab_bench.py $m --gpus 1 --steps 20 --warmup 5 > $O/$m.json 2> $O/$m.err; echo "$m rc=$?"
  python - <<PY
import json
try:
    r = json.loads(open("$O/$m.json").read().strip().splitlines()[-1])
    t = r["tinyllama_checkpoint"]; a = t["api"]
    print("$m", r["value"], a["ms_both"], a["ms_host_until_compress_model_returns"], a["ms_host_until_decompress_model_returns"], t["ms_whole_checkpoint_one_launch_per_module"],
          r["bitmask"].get("api_compress_us"), r["marlin24"].get("compress_us_default"))
except Exception as e:
    print("$m", "ERR", repr(e)[:200])
PY
done
