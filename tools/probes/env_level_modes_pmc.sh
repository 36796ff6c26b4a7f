#!/bin/bash
# counters of the env level per engine (= per workspace allocation), to see what the slow allocations have more of
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/envmodes_$i
  N_ENGINES=6 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/envmodes_$i -- python3 $R/tools/probes/env_level_modes.py > $R/gpurun_out/envmodes_$i.out 2> $R/gpurun_out/envmodes_$i.err
  tail -1 $R/gpurun_out/envmodes_$i.out | cut -c1-1200
  python3 - <<PY
import csv, glob, collections
fs = glob.glob("$R/gpurun_out/envmodes_$i/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file"); raise SystemExit
rows = [r for r in csv.DictReader(open(fs[0])) if "k_rb_level<3" in r["Kernel_Name"] or "k_rb_levelILi3" in r["Kernel_Name"]]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
per = len(ids) // 6
eng = {d: min(k // per, 5) for k, d in enumerate(ids)}
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    acc[eng[int(r["Dispatch_Id"])]][r["Counter_Name"]] += float(r["Counter_Value"])
for e in sorted(acc):
    print("engine", e, {k: round(v / per, 1) for k, v in acc[e].items()})
PY
done
