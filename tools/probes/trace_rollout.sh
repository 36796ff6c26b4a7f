R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trr
rocprofv3 --kernel-trace --output-format csv -d $O/trr -- python3 $R/tools/bench_rollout.py > /dev/null 2> $O/trr.err
python3 $R/tools/prof_trace.py $(find $O/trr -name "*kernel_trace.csv") 24
rm -rf $O/trr
