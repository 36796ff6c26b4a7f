#!/bin/bash
# workspace placement trial (IGI_WS_TRIALS, teacher_native.py) on / off: engines in one process, then alternating bench runs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
for t in 4 1 4 1; do echo "IGI_WS_TRIALS=$t"; IGI_WS_TRIALS=$t python3 tools/probes/env_level_modes.py 2>&1 | tail -1 | cut -c1-1400; done
for v in 4 1 4 1 4 1 4 1 4 1; do
  IGI_WS_TRIALS=$v python3 bench.py --no-cpu-baseline --no-student --no-peak-probe --no-experiments --steps 20 --warmup 3 > $O/r06_bench_ws_$v.json 2> $O/r06_bench_ws_$v.err
  python3 - <<PY
import json
try:
    r = json.loads([l for l in open("$O/r06_bench_ws_$v.json") if l.startswith("{")][-1])
    lv = {l["level"].split(":")[0]: l["avg_us"] for l in r["roofline"].get("levels", []) if l["kernel"] == "k_rb_level"}
    print("TRIALS=$v", r["value"], "updates/s", r["ms_per_step"], "ms;", lv, r["config"].get("workspace_trial_ms"))
except Exception as e:
    print("bench ws=$v failed:", e); print(open("$O/r06_bench_ws_$v.err").read()[-1500:])
PY
done
