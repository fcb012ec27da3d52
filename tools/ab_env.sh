# usage: ab_env.sh VAR v1 v2 ... ; runs bench with VAR=v for streams 1 and 2
var=$1; shift
for v in "$@"; do
  for s in ${STREAMS:-1 2}; do
  env $var=$v timeout 300 python bench.py --no-cpu --streams $s 2>&1 | tail -1 > gpurun_out/b.json; python - <<PY
import json
d=json.load(open("gpurun_out/b.json"))
print("$var=$v streams $s", round(d["value"]), round(d["roofline"]["frac"],3), {k:round(v) for k,v in d["kernel_ms_per_step"].items() if v>20})
PY
  done
done
