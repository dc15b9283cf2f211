#!/bin/bash
# round-4 GPU call I: marlin-24 with the XCDs on adjacent row blocks (CT_M24_X=1) against the shipped mapping
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
for m in 0 1 0 1; do CT_M24_X=$m timeout 300 python tools/exp_r04.py marlin | sed "s/^{/{\"xcd_rows\": $m, /"; done > $O/marlin.jsonl 2> $O/marlin.err; cat $O/marlin.jsonl
cd /tmp; for m in 0 1; do for pass in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pp; CT_M24_X=$m rocprofv3 --kernel-trace --pmc $pass -d /tmp/pp -o run -- python $GRAFT_REPO_ROOT/tools/prof_marlin.py > /dev/null 2>&1; echo "xcd_rows=$m"; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pp/run_results.db | grep -i "marlin24_fused" | cut -c1-160; done; done
