#!/bin/bash
# round-4 GPU call G: the profile passes of the round (kernel trace + FETCH / WRITE / SQ counter passes of the bench command; SQ passes of the bitmask and marlin kernels)
export TMPDIR=/tmp
R=$(pwd)
bash tools/profile_round.sh r04 > gpurun_out/profile_r04.log 2>&1; echo "profile_round rc=$?"
cd $R
H=$(cat compressed_tensors_amd/libct_hip.so.srchash 2>/dev/null)
{ echo "# srchash $H"; echo "# tools/pmc_res.sh: rocprofv3 --kernel-trace --pmc <pass> of ct_bitmask_compress at 8192^2 bf16 50 % (tools/exp_r02.py bmres1), one pass per counter group"; MODES=1 bash tools/pmc_res.sh; } > gpurun_out/profile_r04/bitmask_sq.txt 2>&1
cd $R
{ echo "# srchash $H"; echo "# tools/pmc_m24.sh: rocprofv3 --kernel-trace --pmc <pass> of the marlin-24 leg (tools/bench_leg.py marlin24_leg)"; bash tools/pmc_m24.sh; } > gpurun_out/profile_r04/marlin_sq.txt 2>&1
cd $R
ls -la gpurun_out/profile_r04; head -30 gpurun_out/profile_r04/headline_trace.txt; cat gpurun_out/profile_r04/marlin_sq.txt | cut -c1-160
