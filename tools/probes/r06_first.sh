#!/bin/bash
# round 6, first GPU pass: the x3 experiment, norm-fusion A/B, then the teacher / edge tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 tools/probes/x3_probe.py > $O/r06_x3_probe.json 2> $O/r06_x3_probe.err; echo "x3 rc=$?"; tail -3 $O/r06_x3_probe.err
for nf in 1 0 1 0; do
  IGI_NORM_FUSE=$nf python3 bench.py --no-cpu-baseline --no-student --steps 20 --warmup 3 > $O/r06_bench_nf$nf.json 2> $O/r06_bench_nf$nf.err
  python3 - <<PY
import json
try:
    r = json.loads([l for l in open("$O/r06_bench_nf$nf.json") if l.startswith("{")][-1])
    print("NORM_FUSE=$nf", r["value"], "updates/s", r["ms_per_step"], "ms;", {k["name"]: k["avg_us"] for k in r["kernels"][:14]})
    print("   peaks:", {k: v for k, v in r["roofline"].items() if k.startswith("peak_") or k.startswith("gemm_kloop")})
except Exception as e:
    print("bench nf=$nf failed:", e); print(open("$O/r06_bench_nf$nf.err").read()[-1500:])
PY
done
timeout 1500 python3 -m pytest tests/test_gpu_teacher.py tests/test_gpu_edges.py tests/test_gpu_dp.py::test_native_rccl_update_on_a_one_rank_communicator -x -q 2>&1 | tail -15
