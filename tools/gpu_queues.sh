#!/bin/bash
# development: chunk-stream count against the number of hardware queues HIP may use (GPU_MAX_HW_QUEUES, default 4)
for q in 4 8; do for s in 4 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --ab --steps 2 --streams $s 2>gpurun_out/q.err | tail -1 > gpurun_out/q.json
  python - "$q" "$s" <<'PY'
import json,sys
try:
    j=json.load(open("gpurun_out/q.json")); k=j["kernel_ms_per_step"]
    print(f"hwq={sys.argv[1]} streams={sys.argv[2]}: {j['value']:9.0f} win/s chunk={j['config']['chunk_windows']} " + " ".join(f"{a[2:]}={b:.0f}" for a,b in k.items() if b>=50))
except Exception as e:
    print(sys.argv[1:], "FAILED", e, open("gpurun_out/q.err").read()[-600:])
PY
done; done
