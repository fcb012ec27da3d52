#!/bin/bash
# development: A/B several library builds on the GPU box: tools/gpu_ab.sh "bench args" name1 name2 ...
# prints windows/s and the per-kernel milliseconds of each variant (variants from tools/build_variant.sh; "main" = the product build)
args=$1; shift
for v in "$@"; do
  lib=vechat_amd/lib/variants/libvechat_hip_$v.so
  [ "$v" = main ] && lib=vechat_amd/lib/libvechat_hip.so
  VECHAT_HIP_LIB=$lib python bench.py --no-cpu --ab $args 2>gpurun_out/ab_$v.err | tail -1 > gpurun_out/ab_$v.json
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
try:
    j=json.load(open(f"gpurun_out/ab_{v}.json"))
    k=j["kernel_ms_per_step"]
    print(f"{v:>10}: {j['value']:9.0f} win/s  " + " ".join(f"{a[2:]}={b:.0f}" for a,b in k.items() if b>=1))
except Exception as e:
    print(v, "FAILED", e, open(f"gpurun_out/ab_{v}.err").read()[-500:])
PY
done
