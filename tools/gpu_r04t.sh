#!/bin/bash
# round 4, run T: the bench of the tree of commit 0afe0a1 (ab_old/, same kernel library) against the current tree on ONE lease, twice each
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04t; mkdir -p $O
for i in 1 2; do
  (cd ab_old && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/old$i.json 2> $O/old$i.err); echo "old rc=$?"
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/new$i.json 2> $O/new$i.err; echo "new rc=$?"
done
python - <<'PY'
import json
for n in ("old1", "new1", "old2", "new2"):
    try:
        r = json.loads(open(f"gpurun_out/r04t/{n}.json").read().strip().splitlines()[-1])
        t = r["tinyllama_checkpoint"]; a = t["api"]
        print(n, r["value"], a["ms_both"], a["ms_host_until_compress_model_returns"], a["ms_host_until_decompress_model_returns"], t["ms_whole_checkpoint_one_launch_per_module"],
              r["cpu_baseline"]["value"], r["bitmask"].get("api_compress_us"), r["marlin24"].get("compress_us_default"))
    except Exception as e:
        print(n, "ERR", repr(e)[:200])
PY
