#!/bin/bash
# round 4, run S: is the ModelCompressor figure of the bench (1.20 ms in runs Q / R, 1.11 ms in run P) the box or the code?
export TMPDIR=/tmp
O=gpurun_out/r04s; mkdir -p $O
timeout 300 python tools/exp_r04.py hostmodel > $O/hostmodel.json 2> $O/hostmodel.err; echo "hostmodel rc=$?"; cat $O/hostmodel.json
timeout 400 python tools/exp_r04.py hostab > $O/hostab.jsonl 2> $O/hostab.err; echo "hostab rc=$?"; cat $O/hostab.jsonl; tail -3 $O/hostab.err
for i in 1 2; do
  timeout 300 python tools/bench_leg.py tinyllama_leg > $O/leg$i.json 2> $O/leg$i.err; echo "leg rc=$?"
  python - <<PY
import json
r=json.loads(open("$O/leg$i.json").read().strip().splitlines()[-1])["tinyllama_leg"]
print({k:v for k,v in r["api"].items() if k.startswith("ms_") or k.startswith("api_")}, r.get("ms_whole_checkpoint_one_launch_per_module"))
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04s/bench.json").read().strip().splitlines()[-1])
print("value", r["value"], "frac", r["roofline"]["frac"])
print({k:v for k,v in r["tinyllama_checkpoint"]["api"].items() if k.startswith("ms_") or k.startswith("api_")}, r["tinyllama_checkpoint"].get("ms_whole_checkpoint_one_launch_per_module"))
print("cpu_baseline", r["cpu_baseline"]["value"])
PY
