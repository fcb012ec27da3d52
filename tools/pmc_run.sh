# rocprofv3 PMC passes over the throughput probe (one counter group per pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc$i -- python $R/tools/gpu_scale.py 4096 64 500 4096 1 > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $grp"; tail -3 /tmp/pmc$i.log | head -1 | cut -c1-160
  python $R/tools/pmc_summary.py $f | grep "k_fwd\|k_tracew" | cut -c1-400
done
