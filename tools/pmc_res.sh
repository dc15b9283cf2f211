# dev helper (run on the GPU box): kernel-trace + SQ counter passes of the bitmask compress at 8192^2 (tools/exp_r02.py bmres1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in ${MODES:-1}; do
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pp; CT_BITMASK_RESIDENT=$mode rocprofv3 --kernel-trace --pmc $pass -d /tmp/pp -o run -- python $R/tools/exp_r02.py bmres1 > /dev/null 2>&1
  echo "== CT_BITMASK_RESIDENT=$mode"; python $R/tools/prof_summary.py /tmp/pp/run_results.db | grep -i "flat16" 
done; done
