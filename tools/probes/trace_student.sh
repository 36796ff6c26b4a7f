R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trs
rocprofv3 --kernel-trace --output-format csv -d $O/trs -- python3 $R/tools/bench_student.py --config ${1:-3} --updates 1 > /dev/null 2> $O/trs.err
python3 $R/tools/prof_trace.py $(find $O/trs -name "*kernel_trace.csv") 16
rm -rf $O/trs
