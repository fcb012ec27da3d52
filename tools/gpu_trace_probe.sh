#!/bin/bash
# development: per-launch durations of the backtrack kernels by grid size (build phase vs re-alignment), one stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/kt_$v
  VC_TRACE_IMPL=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -- python $R/tools/gpu_scale.py 8192 64 500 8192 1 > /tmp/kt_$v.log 2>&1
  echo "== VC_TRACE_IMPL=$v"; grep "^rep 1" /tmp/kt_$v.log | cut -c1-420
  f=$(find /tmp/kt_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/ktrace_summary.py $f 14
done
