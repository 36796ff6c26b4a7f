#!/bin/bash
# Regenerates the judged profile summaries on the GPU box (run through gpurun from the repository root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r04'
# then copy the gpurun_out/<tag>_* summaries into profiles/ (tracked).  rocprofv3 runs the program itself after `--`
# (python3 ...); counters are collected in their own passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE
# do not fit one pass) and only for our kernels (--kernel-include-regex: with counters armed on every ATen kernel the
# student run segfaulted inside the profiler at an at::native::floor_divide dispatch in round 2).
# Every step checks its own outcome: a failed pass leaves NO summary file behind (round 2 committed a 0-byte JSON).
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAILED=""
ONLY='igi::|gemm_dma|k_[a-z_]+'
# bench.py's workspace choice (TeacherEngine.tune_workspace): one warm-up update + one per candidate, before everything else
WS_TRIALS=${IGI_WS_TRIALS:-6}
WS_SKIP=0; [ "$WS_TRIALS" -gt 1 ] && WS_SKIP=$((WS_TRIALS + 1))

run() {   # run <name> <log> -- cmd...   : records a failure, keeps going
  local name=$1 log=$2; shift 3
  if ! "$@" > "$log.out" 2> "$log.err"; then
    echo "[collect] FAILED: $name (see $log.err)"; tail -5 "$log.err"; FAILED="$FAILED $name"; return 1
  fi
}

stats_csv() {  # stats_csv <dir> <dest>: the kernel_stats.csv of a --stats pass
  local f; f=$(find "$1" -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ] || [ ! -s "$f" ]; then echo "[collect] no kernel_stats.csv under $1"; FAILED="$FAILED stats:$1"; return 1; fi
  cp "$f" "$2"
}

traffic() {    # traffic <fetch dir> <write dir> <dest json>
  local f w; f=$(find "$1" -name "*counter_collection.csv" | head -1); w=$(find "$2" -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ] || [ -z "$w" ]; then echo "[collect] missing counter_collection.csv ($1 / $2)"; FAILED="$FAILED pmc:$3"; return 1; fi
  if ! python3 $R/tools/hbm_traffic.py "$f" "$w" > "$3.tmp" 2> "$3.err" || [ ! -s "$3.tmp" ]; then
    echo "[collect] hbm_traffic.py failed for $3"; cat "$3.err"; rm -f "$3.tmp"; FAILED="$FAILED traffic:$3"; return 1
  fi
  mv "$3.tmp" "$3"; rm -f "$3.err"
}

# ---- teacher bench: kernel trace + stats (the same command the driver runs, minus the CPU / student legs)
rm -rf $O/${TAG}_stats
run bench_stats $O/${TAG}_bench_under_rocprof -- rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- \
    python3 $R/bench.py --no-cpu-baseline --no-student --no-experiments --no-peak-probe \
  && cp $O/${TAG}_bench_under_rocprof.out $O/${TAG}_bench_under_rocprof.json \
  && stats_csv $O/${TAG}_stats $O/${TAG}_bench_kernel_stats.csv \
  && python3 $R/tools/levels_from_trace.py "$(find $O/${TAG}_stats -name '*kernel_trace.csv' | head -1)" --skip-first-updates $WS_SKIP > $O/${TAG}_bench_levels.csv
run bench_fetch $O/${TAG}_pmc_fetch -- rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$ONLY" --output-format csv \
    -d $O/${TAG}_pmc_fetch_d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-student --no-experiments --no-roofline
run bench_write $O/${TAG}_pmc_write -- rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "$ONLY" \
    --output-format csv -d $O/${TAG}_pmc_write_d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-student --no-experiments --no-roofline
traffic $O/${TAG}_pmc_fetch_d $O/${TAG}_pmc_write_d $O/${TAG}_hbm_traffic.json

# ---- student: configs[2] (tactile + lin, 2048 envs) and the configs[3] share (tactile + PointNet x2 + lin, 512 envs)
for C in 3 4; do
  ENVS=2048; [ $C = 4 ] && ENVS=512
  rm -rf $O/${TAG}_student_c${C}_stats
  run student_c${C}_stats $O/${TAG}_student_c${C}_under_rocprof -- rocprofv3 --kernel-trace --stats --output-format csv \
      -d $O/${TAG}_student_c${C}_stats -- python3 $R/tools/bench_student.py --config $C --envs $ENVS --updates 1 \
    && cp $O/${TAG}_student_c${C}_under_rocprof.out $O/${TAG}_student_c${C}_under_rocprof.json \
    && stats_csv $O/${TAG}_student_c${C}_stats $O/${TAG}_student_c${C}_kernel_stats.csv
  run student_c${C}_fetch $O/${TAG}_student_c${C}_pmc_fetch -- rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$ONLY" \
      --output-format csv -d $O/${TAG}_student_c${C}_pmc_fetch_d -- python3 $R/tools/bench_student.py --config $C --envs $ENVS --updates 1
  run student_c${C}_write $O/${TAG}_student_c${C}_pmc_write -- rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace \
      --kernel-include-regex "$ONLY" --output-format csv -d $O/${TAG}_student_c${C}_pmc_write_d -- \
      python3 $R/tools/bench_student.py --config $C --envs $ENVS --updates 1
  traffic $O/${TAG}_student_c${C}_pmc_fetch_d $O/${TAG}_student_c${C}_pmc_write_d $O/${TAG}_student_c${C}_hbm_traffic.json
done

# raw per-dispatch traces are large: keep the summaries only
rm -rf $O/${TAG}_pmc_fetch_d $O/${TAG}_pmc_write_d $O/${TAG}_student_c*_pmc_fetch_d $O/${TAG}_student_c*_pmc_write_d
find $O/${TAG}_stats $O/${TAG}_student_c*_stats -name "*kernel_trace*" -delete 2>/dev/null
find $O -name "*agent_info*" -delete 2>/dev/null
# every summary carries the hash of the sources the profiled library was built from (bench.py refuses profile-sourced
# roofline fields from another build)
python3 $R/tools/stamp_profiles.py $O $TAG || FAILED="$FAILED stamp"
ls -la $O/${TAG}_*.json $O/${TAG}_*.csv 2>/dev/null
if [ -n "$FAILED" ]; then echo "[collect] FAILED STEPS:$FAILED"; exit 1; fi
echo "[collect] all passes ok"
