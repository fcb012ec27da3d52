#!/bin/bash
# k_tracew with 8 lanes x 8 alignments per wave (default) against 16 x 4 (VC_TRACE_TL=16): parity tests on both, then config C and
# config C with partial-span layers.
for tl in 8 16; do
  export VC_TRACE_TL=$tl
  echo "== VC_TRACE_TL=$tl"
  python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
  for fp in 0 0.2; do python tools/gpu_scale.py 32768 64 500 0 4 $fp 2>&1 | grep "rep 1" | cut -c1-420; done
done
unset VC_TRACE_TL
python tools/gpu_stress.py 160 11 2>&1 | tail -2
