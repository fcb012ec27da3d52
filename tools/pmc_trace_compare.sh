#!/bin/bash
# development: dynamic instruction counts of the backtrack implementations (VC_TRACE_IMPL) by PMC
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/pmct_$v
  VC_TRACE_IMPL=$v timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/pmct_$v -- python $R/tools/gpu_scale.py 4096 64 500 4096 1 > /tmp/pmct_$v.log 2>&1
  f=$(find /tmp/pmct_$v -name "*counter_collection.csv" | head -1)
  echo "== VC_TRACE_IMPL=$v"; grep "^rep 1" /tmp/pmct_$v.log | cut -c1-200
  python $R/tools/pmc_summary.py $f | grep "k_trace\|k_rows" | cut -c1-400
done
