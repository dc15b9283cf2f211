#!/bin/bash
# round 4, run V: old tree / new tree without the CPU-baseline leg in front, new tree with the host extension imported late
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04v; mkdir -p $O
run() {  # name, directory, mode
  (cd $2 && timeout 400 python tools/ab_bench.py $3 --gpus 1 --steps 20 --warmup 5 > $O/$1.json 2> $O/$1.err); echo "$1 rc=$?"
  python - <<PY
import json
try:
    r = json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    t = r["tinyllama_checkpoint"]; a = t["api"]
    print("$1", r["value"], a["ms_both"], a["ms_host_until_compress_model_returns"], a["ms_host_until_decompress_model_returns"], t["ms_whole_checkpoint_one_launch_per_module"],
          r["bitmask"].get("api_compress_us"), r["marlin24"].get("compress_us_default"))
except Exception as e:
    print("$1", "ERR", repr(e)[:200])
PY
}
run old_nocpu ab_old nocpu
run new_nocpu . nocpu
run new_lateimport . lateimport
run old_asis ab_old asis
run new_asis . asis
