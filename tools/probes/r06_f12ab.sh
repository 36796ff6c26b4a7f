#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout 600 python3 -m pytest tests/test_gpu_teacher.py -x -q -k "golden or bitwise or stepwise or infer" 2>&1 | tail -4
for v in 1 0 1 0; do
  IGI_FWD12=$v python3 bench.py --no-cpu-baseline --no-student --no-peak-probe --steps 20 --warmup 3 > $O/r06_bench_f12_$v.json 2> $O/r06_bench_f12_$v.err
  python3 - <<PY
import json
try:
    r = json.loads([l for l in open("$O/r06_bench_f12_$v.json") if l.startswith("{")][-1])
    print("FWD12=$v", r["value"], "updates/s", r["ms_per_step"], "ms;", {k["name"]: k["avg_us"] for k in r["kernels"][:12]})
except Exception as e:
    print("bench f12=$v failed:", e); print(open("$O/r06_bench_f12_$v.err").read()[-1500:])
PY
done
