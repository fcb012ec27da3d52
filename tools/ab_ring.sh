# usage: ab_ring.sh r1 r2 ...   (libs prebuilt as vechat_amd/lib/libvechat_hip.so.ringN)
for r in "$@"; do
  cp vechat_amd/lib/libvechat_hip.so.ring$r vechat_amd/lib/libvechat_hip.so
  for s in 1 2; do
  timeout 300 python bench.py --no-cpu --streams $s 2>&1 | tail -1 > gpurun_out/b.json; python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("ring $r streams $s", round(d["value"]), round(d["roofline"]["frac"],3), {k:round(v) for k,v in d["kernel_ms_per_step"].items() if v>20})
PY
  done
done
