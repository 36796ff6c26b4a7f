#!/bin/bash
# Regenerates the judged profile summaries on the GPU box (run through gpurun from the repository root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
# then copy gpurun_out/r2_stats/*/*_kernel_stats.csv, gpurun_out/r02_*.json ... into profiles/ (see the end of DESIGN.md 7).
# rocprofv3 runs the program itself after `--` (python3 ...), counters in their own passes (MI355X_MICROARCH.md, HBM section).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_stats -- python3 $R/bench.py --no-cpu-baseline --no-student > $O/r2_bench_under_rocprof.json 2> $O/r2_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r2_pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-student --no-roofline > /dev/null 2> $O/r2_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/r2_pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-student --no-roofline > /dev/null 2> $O/r2_pmc_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2_student_stats -- python3 $R/tools/bench_student.py --config 3 --updates 1 > $O/r2_student_c3_under_rocprof.json 2> $O/r2_student_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r2_student_pmc_fetch -- python3 $R/tools/bench_student.py --config 3 --updates 1 > /dev/null 2> $O/r2_student_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/r2_student_pmc_write -- python3 $R/tools/bench_student.py --config 3 --updates 1 > /dev/null 2> $O/r2_student_pmc_write.err
python3 $R/tools/hbm_traffic.py $(find $O/r2_pmc_fetch -name "*counter_collection.csv") $(find $O/r2_pmc_write -name "*counter_collection.csv") > $O/r02_hbm_traffic.json
python3 $R/tools/hbm_traffic.py $(find $O/r2_student_pmc_fetch -name "*counter_collection.csv") $(find $O/r2_student_pmc_write -name "*counter_collection.csv") > $O/r02_student_c3_hbm_traffic.json
rm -rf $O/r2_pmc_fetch $O/r2_pmc_write $O/r2_student_pmc_fetch $O/r2_student_pmc_write
find $O/r2_stats $O/r2_student_stats -name "*kernel_trace*" -delete
find $O -name "*agent_info*" -newer $R/bench.py -delete
du -sh $O/r2_* | tail -12
