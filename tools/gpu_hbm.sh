#!/bin/bash
# development: HBM bytes per kernel class (rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE) of the throughput probe
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$grp
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$grp -- python $R/tools/gpu_scale.py ${1:-8192} 64 500 ${2:-8192} ${3:-1} > /tmp/pmc_$grp.log 2>&1
  f=$(find /tmp/pmc_$grp -name "*counter_collection.csv" | head -1)
  echo "== $grp"; grep "^rep 1" /tmp/pmc_$grp.log | cut -c1-120
  python $R/tools/pmc_summary.py $f | cut -c1-300
done
