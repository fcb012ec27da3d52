for cfg in "2 8192" "3 8192" "4 8192" "3 4096" "4 4096" "2 16384"; do set -- $cfg
timeout 300 python bench.py --no-cpu --streams $1 --chunk $2 2>&1 | tail -1 > gpurun_out/b.json; python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("$cfg", round(d["value"]), round(d["roofline"]["frac"],3), {k:round(v) for k,v in d["kernel_ms_per_step"].items() if v>20})
PY
done
