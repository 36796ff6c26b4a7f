#!/bin/bash
# the round's stand-alone measurements once more on the FINAL build (so that their build stamp is true)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
for v in 1 0 1 0; do
  IGI_POLICY_FUSED=$v python3 tools/bench_rollout.py 2>/dev/null | tail -1 > $O/f_rollout_${v}_$RANDOM.json
done
cat $O/f_rollout_1_*.json > $O/f_rollout_fused1.txt; cat $O/f_rollout_0_*.json > $O/f_rollout_fused0.txt; rm -f $O/f_rollout_?_*.json
python3 tools/probes/policy_fwd_time.py > $O/f_policy_fwd_time.json 2>/dev/null
python3 tools/probes/pointnet_bench.py > $O/f_pointnet.json 2>/dev/null
IGI_PN_COLMAX=1 python3 tools/probes/pointnet_bench.py > $O/f_pointnet_colmax.json 2>/dev/null
PN_SQ_OUT=gpurun_out/f_pointnet_sq_counters.json bash tools/probes/pointnet_sq.sh > /dev/null 2>&1
IGI_PN_COLMAX=1 PN_SQ_OUT=gpurun_out/f_pointnet_colmax_sq_counters.json bash tools/probes/pointnet_sq.sh > /dev/null 2>&1
cd $R
python3 tools/probes/x3_probe.py > $O/f_x3_probe.json 2>/dev/null
for v in 1 0; do IGI_PCL_ONE_LAUNCH=$v python3 tools/bench_student.py --config 4 --envs 512 --updates 3 2>/dev/null | tail -1 > $O/f_student_c4_pcl$v.json; done
ls -la $O/f_*
