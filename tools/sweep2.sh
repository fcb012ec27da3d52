for lib in ring6c ring8; do cp vechat_amd/lib/libvechat_hip.so.$lib vechat_amd/lib/libvechat_hip.so
for cfg in "2 4096" "3 4096" "2 3072" "3 3072"; do set -- $cfg
timeout 300 python bench.py --no-cpu --streams $1 --chunk $2 2>&1 | tail -1 > gpurun_out/b.json; python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("$lib $cfg", round(d["value"]), round(d["roofline"]["frac"],3), {k:round(v) for k,v in d["kernel_ms_per_step"].items() if v>20})
PY
done; done
cp vechat_amd/lib/libvechat_hip.so.ring8 vechat_amd/lib/libvechat_hip.so
