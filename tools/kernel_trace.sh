#!/bin/bash
# per-launch kernel durations of the bench, aggregated by (kernel symbol, grid): gpurun -- 'bash tools/kernel_trace.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-student --no-roofline > /dev/null 2> $O/tr.err
python3 $R/tools/prof_trace.py $(find $O/tr -name "*kernel_trace.csv") 30
rm -rf $O/tr
