#!/bin/bash
# the round's headline records on the final build (what tools/probes/closing.sh writes last), after the host-side change
# that followed the closing pass (TeacherEngine.tune_workspace)
cd ${GRAFT_REPO_ROOT:-.}
IGI_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2> gpurun_out/r06_bench_2rank.err | grep "^{" | tail -1 > gpurun_out/r06_bench_2rank_gloo_one_gpu.json
python3 bench.py 2> gpurun_out/r06_bench.err | tail -1 > gpurun_out/r06_bench_n1.json
cut -c1-400 gpurun_out/r06_bench_n1.json; tail -3 gpurun_out/r06_bench.err | cut -c1-300
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-200
python3 tools/stamp_profiles.py gpurun_out r06 | tail -1
