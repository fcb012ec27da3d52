#!/bin/bash
# k_tracew deciding run-ending vertical / horizontal steps from the speculated lanes (default) against the general step for every
# run end (VC_TRACE_NO_EXTRA=1): parity first, then config C and config C with partial-span layers.
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/gpu_stress.py 160 11 2>&1 | tail -2
VC_PIPE=1 python tools/gpu_stress.py 60 5 2>&1 | tail -1
for ne in "" 1 "" 1; do
  if [ -n "$ne" ]; then export VC_TRACE_NO_EXTRA=1; else unset VC_TRACE_NO_EXTRA; fi
  for fp in 0 0.2; do echo -n "NO_EXTRA=${ne:-0} frac_partial $fp: "; python tools/gpu_scale.py 100000 64 500 0 0 $fp 2>&1 | grep "rep 1" | grep -o "rep 1: [0-9.]* win/s\|rounds=[0-9/]*\|.k_trace.: [0-9.]*\|.k_fwd.: [0-9.]*" | tr "\n" " "; echo; done
done
