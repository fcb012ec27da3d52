#!/bin/bash
# development: kernel trace of config C (args: windows) and the per-stream time split
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ktl
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -- python $R/tools/gpu_scale.py ${1:-100000} 64 500 > /tmp/ktl.log 2>&1
grep "^rep" /tmp/ktl.log | cut -c1-120
f=$(find /tmp/ktl -name "*kernel_trace.csv" | head -1)
python $R/tools/stream_path.py $f
python $R/tools/timeline2.py $f | head -40
