#!/bin/bash
# round 4, run W: per-function host time of the model path in the bench, host extension imported early (slow state) vs late
export TMPDIR=/tmp
O=gpurun_out/r04w; mkdir -p $O
for m in timed timed_late; do
  timeout 400 python tools/ab_bench.py $m --gpus 1 --steps 20 --warmup 5 > $O/$m.json 2> $O/$m.err; echo "$m rc=$?"; tail -1 $O/$m.err
  python - <<PY
import json
r = json.loads(open("$O/$m.json").read().strip().splitlines()[-1])
a = r["tinyllama_checkpoint"]["api"]
print("$m", a["ms_both"], a["ms_host_until_compress_model_returns"], a["ms_host_until_decompress_model_returns"])
PY
done
