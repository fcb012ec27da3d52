#!/bin/bash
# development: kernel timeline of the throughput probe; args: windows chunk streams
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktl
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -- python $R/tools/gpu_scale.py ${1:-32768} 64 500 ${2:-0} ${3:-0} > /tmp/ktl.log 2>&1
grep "^rep" /tmp/ktl.log | cut -c1-200
f=$(find /tmp/ktl -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline2.py $f | head -${4:-60}
python $R/tools/ktrace_summary.py $f 24
