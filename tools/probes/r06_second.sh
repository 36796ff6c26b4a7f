#!/bin/bash
# round 6, second GPU pass: the multi-object PointNet, the student trajectory at configs[2], student A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_pointnet.py tests/test_gpu_ops.py tests/test_gpu_glue.py -x -q 2>&1 | tail -8
for v in 1 0 1 0; do
  IGI_PCL_ONE_LAUNCH=$v python3 tools/bench_student.py --config 4 --envs 512 --updates 3 > $O/r06_student_c4_pcl$v.json 2> $O/r06_student_c4_pcl$v.err
  python3 - <<PY
import json
try:
    r = json.loads([l for l in open("$O/r06_student_c4_pcl$v.json") if l.startswith("{")][-1])
    print("PCL_ONE_LAUNCH=$v", r["ms_per_optimizer_step"], "ms/step;", {k["name"]: (k["launches_per_update"], k["avg_us"]) for k in r["native_kernels"][:8]})
except Exception as e:
    print("student failed:", e); print(open("$O/r06_student_c4_pcl$v.err").read()[-1500:])
PY
done
timeout 1700 python3 -m pytest tests/test_gpu_student_scale.py -x -q -k "trajectory" 2>&1 | tail -15
timeout 900 python3 -m pytest tests/test_gpu_student.py tests/test_gpu_dp.py -x -q 2>&1 | tail -8
