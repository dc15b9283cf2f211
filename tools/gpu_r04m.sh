#!/bin/bash
# round-4 GPU call M: smoke + the module-path tests + one bench line with the final host extension
O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q -x -k "model or batch or module or dropin or upstream" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r=json.load(open("gpurun_out/r04m/bench.json"))
a=r["tinyllama_checkpoint"]["api"]; print("value", r["value"], "api", a.get("ms_both"), a.get("api_over_kernels"), a.get("error"))
PY
