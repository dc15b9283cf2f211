#!/bin/bash
# round-4 GPU call E: bitmask early poll with the bitmask stores before / after the hand-off, marlin (scale_packed from LDS, early permutation row), chunked model launches
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
for late in 0 1 0 1; do CT_BM_LATE=$late timeout 200 python tools/exp_r04.py bmx | sed "s/^{/{\"late\": $late, /"; done > $O/bmx.jsonl 2> $O/bmx.err; cat $O/bmx.jsonl
CT_BITMASK_RESIDENT=3 CT_BM_LATE=0 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_late0.json 2>> $O/bmx.err
CT_BITMASK_RESIDENT=3 CT_BM_LATE=1 timeout 200 python tools/exp_r04.py bmstamps > $O/stamps_late1.json 2>> $O/bmx.err
timeout 300 python tools/exp_r04.py marlin > $O/marlin.json 2> $O/marlin.err; cat $O/marlin.json
timeout 900 python -m pytest tests -m gpu -q -x -k "bitmask or marlin or model or batch or sparse" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04e/bench.json"))
    print("value", r["value"], "frac", r["roofline"]["frac"])
    for k in ("bitmask","marlin24","tinyllama_checkpoint"):
        v=r.get(k,{})
        print(k, {kk:vv for kk,vv in v.items() if any(t in kk for t in ("api","_us","ms_","error"))})
except Exception as e:
    print("bench parse failed", e)
PY
