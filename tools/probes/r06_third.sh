#!/bin/bash
# round 6, third GPU pass: the persistent policy kernel (tests + A/B), PointNet bench default vs COLMAX with SQ counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_rollout.py tests/test_gpu_ppo_api.py tests/test_gpu_learning.py tests/test_gpu_train_entry.py -x -q 2>&1 | tail -12
for v in 1 0 1 0; do
  IGI_POLICY_FUSED=$v python3 tools/bench_rollout.py > $O/r06_rollout_fused$v.json 2> $O/r06_rollout_fused$v.err || tail -5 $O/r06_rollout_fused$v.err
  echo "POLICY_FUSED=$v $(tail -1 $O/r06_rollout_fused$v.json)"
done
python3 tools/probes/pointnet_bench.py > $O/r06_pointnet.json 2> $O/r06_pointnet.err; echo "pointnet default:"; cat $O/r06_pointnet.json | tr -d '\n '; echo
IGI_PN_COLMAX=1 python3 tools/probes/pointnet_bench.py > $O/r06_pointnet_colmax.json 2> $O/r06_pointnet_colmax.err; echo "pointnet colmax:"; cat $O/r06_pointnet_colmax.json | tr -d '\n '; echo
PN_SQ_OUT=gpurun_out/r06_pointnet_sq_counters.json bash tools/probes/pointnet_sq.sh 2>&1 | tail -2
IGI_PN_COLMAX=1 PN_SQ_OUT=gpurun_out/r06_pointnet_colmax_sq_counters.json bash tools/probes/pointnet_sq.sh 2>&1 | tail -2
