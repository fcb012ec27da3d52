#!/bin/bash
# development: alternate runs of library variants / environments on one box.  usage: tools/gpu_ab2.sh OUTDIR "label|ENV=..|lib" ...   (lib: main or a variant name)
O=$1; shift; mkdir -p $O
for rep in 1 2; do
  for spec in "$@"; do
    IFS='|' read -r label envs lib <<< "$spec"
    path=vechat_amd/lib/libvechat_hip.so; [ "$lib" != "main" ] && path=vechat_amd/lib/variants/libvechat_hip_$lib.so
    env $envs VECHAT_HIP_LIB=$path timeout 300 python bench.py --no-cpu --no-extras --steps 3 > $O/ab_${label}_$rep.json 2> $O/ab_${label}_$rep.err
    python - "$label" "$rep" "$O/ab_${label}_$rep.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    k = j["roofline"]["k_fwd"]
    print(f"{sys.argv[1]:>14} rep {sys.argv[2]}: {j['value']:8.0f} windows/s  {j['ms_per_step']:7.1f} ms/step  k_fwd launches {k['launches_per_step']:.0f} avg {k['avg_launch_ms']:.3f} ms busy {k['busy_ms_per_step']:.0f} ms", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
  done
done
