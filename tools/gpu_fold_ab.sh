#!/bin/bash
# Partial-span layers put sequences of many lengths into one layer: one folded k_fwd launch per layer (default) against one launch
# per width class (VC_NO_FOLD=1), on config C with 20 % partial-span layers and on plain config C.
for fp in 0.2 0; do
  for nf in "" 1; do
    echo "== frac_partial $fp VC_NO_FOLD=${nf:-0}"
    if [ -n "$nf" ]; then export VC_NO_FOLD=1; else unset VC_NO_FOLD; fi
    python tools/gpu_scale.py 32768 64 500 0 4 $fp 2>&1 | grep "rep 1" | cut -c1-330
  done
done
