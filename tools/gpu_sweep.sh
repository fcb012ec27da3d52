#!/bin/bash
# development: chunk size / stream count sweep of the bench (args: "chunk:streams" ...), optional env prefix in $SWEEP_ENV
for cs in "$@"; do
  c=${cs%%:*}; s=${cs##*:}
  env $SWEEP_ENV python bench.py --no-cpu --ab --steps 2 --chunk $c --streams $s 2>gpurun_out/sw.err | tail -1 > gpurun_out/sw.json
  python - "$c" "$s" <<'PY'
import json,sys
try:
    j=json.load(open("gpurun_out/sw.json")); k=j["kernel_ms_per_step"]
    print(f"chunk={sys.argv[1]} streams={sys.argv[2]}: {j['value']:9.0f} win/s not_ok={j['windows_not_ok']} chunk_used={j['config']['chunk_windows']} " + " ".join(f"{a[2:]}={b:.0f}" for a,b in k.items() if b>=50))
except Exception as e:
    print(sys.argv[1:], "FAILED", e, open("gpurun_out/sw.err").read()[-600:])
PY
done
