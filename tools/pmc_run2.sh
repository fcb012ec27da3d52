# second set of rocprofv3 PMC passes: what a k_fwd wave waits for (branches, scalar/LDS issue, store-path back-pressure)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_IFETCH" "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmcb$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcb$i -- python $R/tools/gpu_scale.py 4096 64 500 4096 1 > /tmp/pmcb$i.log 2>&1
  f=$(find /tmp/pmcb$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $grp"; tail -3 /tmp/pmcb$i.log | head -1 | cut -c1-160
  python $R/tools/pmc_summary.py $f | grep "k_fwd\|k_tracew" | cut -c1-600
done
