for cfg in "1 8192" "1 16384" "1 32768" "2 16384"; do set -- $cfg
timeout 300 python bench.py --no-cpu --streams $1 --chunk $2 2>&1 | tail -1 > gpurun_out/b.json; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json"))
    print("$cfg", round(d["value"]), round(d["roofline"]["frac"],3), d["config"]["chunk_windows"], {k:round(v) for k,v in d["kernel_ms_per_step"].items() if v>20})
except Exception as e: print("$cfg failed", open("gpurun_out/b.json").read()[:300])
PY
done
