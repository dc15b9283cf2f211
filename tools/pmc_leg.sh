# dev helper (run on the GPU box): kernel-trace + SQ counter passes of ONE bench.py leg (tools/bench_leg.py <leg>), kernels filtered by name
#   usage: bash tools/pmc_leg.sh <leg> <kernel name pattern> > gpurun_out/<tag>/<leg>_sq.txt      e.g.  bitmask_leg "flat16_resident|bitmask_decompress16"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "# srchash $(cat $R/compressed_tensors_amd/libct_hip.so.srchash 2>/dev/null)"
echo "# tools/pmc_leg.sh $1: rocprofv3 --kernel-trace --pmc <pass> of tools/bench_leg.py $1 (kernels matching $2)"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc $pass -d /tmp/pp -o run -- python $R/tools/bench_leg.py $1 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/pp/run_results.db | grep -E -i "$2"
done
