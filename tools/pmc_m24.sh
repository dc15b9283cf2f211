cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace --pmc $pass -d /tmp/pp -o run -- python $R/tools/bench_leg.py marlin24_leg > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/pp/run_results.db | grep -i "marlin24_fused" 
done
