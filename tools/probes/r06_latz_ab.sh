#!/bin/bash
# A/B of the LATZ experiment (rowblock.h MODE 3): parity tests with the switch on, then an alternating bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
IGI_LATZ_FUSE=1 timeout 900 python3 -m pytest tests/test_gpu_teacher.py tests/test_gpu_edges.py -x -q 2>&1 | tail -6
for v in 1 0 1 0; do
  IGI_LATZ_FUSE=$v python3 bench.py --no-cpu-baseline --no-student --no-peak-probe --no-experiments --steps 20 --warmup 3 > $O/r06_bench_latz_$v.json 2> $O/r06_bench_latz_$v.err
  python3 - <<PY
import json
try:
    r = json.loads([l for l in open("$O/r06_bench_latz_$v.json") if l.startswith("{")][-1])
    print("LATZ=$v", r["value"], "updates/s", r["ms_per_step"], "ms;", {k["name"]: k["avg_us"] for k in r["kernels"][:12]})
except Exception as e:
    print("bench latz=$v failed:", e); print(open("$O/r06_bench_latz_$v.err").read()[-1500:])
PY
done
