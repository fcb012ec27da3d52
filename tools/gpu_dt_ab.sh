#!/bin/bash
# development: k_fwd_dt against k_fwd -- kernels alone (one stream) and their PMC instruction counts.   usage: tools/gpu_dt_ab.sh OUTDIR
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6dt}; R=$GRAFT_REPO_ROOT; mkdir -p $O
cd $R
for dt in 1 0; do
  echo "== VC_DT=$dt, one stream, 8192 windows of config C" >> $O/alone.txt
  VC_DT=$dt timeout 300 python tools/gpu_scale.py 8192 64 500 8192 1 2>&1 | grep "^rep" >> $O/alone.txt
done
cd /tmp && export TMPDIR=/tmp
for dt in 1 0; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
    i=$((i+1)); rm -rf /tmp/pmc_dt
    VC_DT=$dt timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_dt -- python $R/tools/gpu_scale.py 4096 64 500 4096 1 > $O/pmc_${dt}_$i.log 2>&1
    fc=$(find /tmp/pmc_dt -name "*counter_collection.csv" | head -1)
    echo "== VC_DT=$dt pass $i: $grp" >> $O/pmc.txt; python $R/tools/pmc_summary.py $fc | grep "k_fwd\|k_tracew\|k_addaln" | cut -c1-500 >> $O/pmc.txt
  done
done
cat $O/alone.txt; cat $O/pmc.txt
