# end-of-round measurement pass: judged summaries -> gpurun_out/ (copied into profiles/ afterwards)
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
bash tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; tail -3 gpurun_out/${TAG}_collect.log
cd $GRAFT_REPO_ROOT
python3 tools/probes/pointnet_bench.py > gpurun_out/${TAG}_pointnet.json 2> gpurun_out/${TAG}_pointnet.err
bash tools/probes/pointnet_sq.sh > /dev/null 2>&1; cp gpurun_out/r03_pointnet_sq_counters.json gpurun_out/${TAG}_pointnet_sq_counters.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 tools/bench_rollout.py 2> gpurun_out/${TAG}_rollout.err | tail -1 > gpurun_out/${TAG}_rollout.json
python3 tools/bench_student.py --config 3 --hw 64 64 2>/dev/null | tail -1 > gpurun_out/${TAG}_student_c3_64x64.json
python3 bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_n1.json
cut -c1-600 gpurun_out/${TAG}_bench_n1.json
