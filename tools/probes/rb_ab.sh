# A/B runs of bench.py on one box: bash tools/probes/rb_ab.sh "name ENV=.. ENV=.." ...   (one quoted argument per configuration)
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-student --steps 10 > gpurun_out/ab_$name.json 2>gpurun_out/ab_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$name.json"))
print("$name", d["value"], d["ms_per_step"], "sum of kernels", round(sum(k["ms_per_update"] for k in d["kernels"]), 3))
for k in d["kernels"][:9]: print("   ", k["name"][:60], k["launches_per_update"], k["avg_us"], k["ms_per_update"])
for l in d["roofline"]["levels"]: print("   L", l["level"][:50], l["avg_us"], l["frac"])
PY
done
