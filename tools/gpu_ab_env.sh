#!/bin/bash
# development: A/B one environment variable on the GPU box: tools/gpu_ab_env.sh VAR "bench args" v1 v2 ...
var=$1; args=$2; shift 2
mkdir -p gpurun_out
for v in "$@"; do
  env $var=$v python bench.py --no-cpu --ab $args 2>gpurun_out/abe_$v.err | tail -1 > gpurun_out/abe_$v.json
  python - "$var=$v" gpurun_out/abe_$v.json gpurun_out/abe_$v.err <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[2]))
    k=j["kernel_ms_per_step"]
    print(f"{sys.argv[1]:>18}: {j['value']:9.0f} win/s not_ok={j['windows_not_ok']} " + " ".join(f"{a[2:]}={b:.0f}" for a,b in k.items() if b>=1))
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[3]).read()[-800:])
PY
done
