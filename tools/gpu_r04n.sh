#!/bin/bash
# round-4 GPU call N: the model-free path with per-tensor D2H events (writer overlaps the copies)
O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_convert_checkpoint.py -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "GBps\|passed\|failed" $O/pytest.log | tail -5
cat gpurun_out/convert_rate.json
