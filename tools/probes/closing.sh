# end-of-round measurement pass: judged summaries -> gpurun_out/ (copied into profiles/ afterwards)
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}
bash tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -3 gpurun_out/${TAG}_collect.log
cd $GRAFT_REPO_ROOT
bash tools/probes/teacher_sq.sh > gpurun_out/${TAG}_teacher_sq.log 2>&1; tail -2 gpurun_out/${TAG}_teacher_sq.log | cut -c1-300
cd $GRAFT_REPO_ROOT
python3 tools/bench_student.py --config 3 --hw 64 64 2>/dev/null | tail -1 > gpurun_out/${TAG}_student_c3_64x64.json
python3 tools/probes/dp_phase_cost.py > gpurun_out/${TAG}_dp_phase_cost.json 2> gpurun_out/${TAG}_dp_phase_cost.err; tail -c 600 gpurun_out/${TAG}_dp_phase_cost.json; echo
IGI_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/${TAG}_bench_2rank.err | grep "^{" | tail -1 > gpurun_out/${TAG}_bench_2rank_gloo_one_gpu.json
python3 - <<PY
import json
try:
    r = json.load(open("gpurun_out/${TAG}_bench_2rank_gloo_one_gpu.json"))
    print("2-rank gloo:", r["value"], r["config"].get("rccl_ranks"), list(r.get("multi_gpu_configs", {}))[:3])
except Exception as e:
    print("2-rank record failed:", e)
PY
python3 bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_n1.json
cut -c1-700 gpurun_out/${TAG}_bench_n1.json
python3 tools/stamp_profiles.py gpurun_out $TAG
