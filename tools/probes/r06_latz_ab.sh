#!/bin/bash
# A/B of the LATZ experiment (rowblock.h MODE 3): parity tests with the switch on, then an alternating bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
# (parity: the full GPU suite runs both settings, tests/test_gpu_teacher.py)
for v in 1 0 1 0 1 0 1 0 1 0; do
  IGI_LATZ_FUSE=$v python3 bench.py --no-cpu-baseline --no-student --no-peak-probe --no-experiments --steps 20 --warmup 3 > $O/r06_bench_latz_$v.json 2> $O/r06_bench_latz_$v.err
  python3 - <<PY
import json
try:
    r = json.loads([l for l in open("$O/r06_bench_latz_$v.json") if l.startswith("{")][-1])
    lv = {l["level"].split(":")[0]: l["avg_us"] for l in r["roofline"].get("levels", []) if l["kernel"] == "k_rb_level"}
    print("LATZ=$v", r["value"], "updates/s", r["ms_per_step"], "ms;", lv, {k["name"]: k["avg_us"] for k in r["kernels"][:12]})
except Exception as e:
    print("bench latz=$v failed:", e); print(open("$O/r06_bench_latz_$v.err").read()[-1500:])
PY
done
