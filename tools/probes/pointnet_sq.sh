# SQ issue-side counters of k_pointnet_fwd (one rocprofv3 --pmc pass over tools/probes/pointnet_bench.py):
# matrix-busy, vector-active and parked cycles per SIMD as fractions of the launch -> $PN_SQ_OUT (default gpurun_out/r06_pointnet_sq_counters.json)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pn_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-include-regex "k_pointnet_fwd" -d gpurun_out/pn_sq -o pn --output-format csv -- python3 tools/probes/pointnet_bench.py > gpurun_out/pn_sq.out 2>&1
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob("gpurun_out/pn_sq/**/*counter_collection.csv", recursive=True)[0]
acc, n = collections.defaultdict(float), 0
for r in csv.DictReader(open(f)):
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
    n += r["Counter_Name"] == "SQ_WAVE_CYCLES"
cyc = acc["GRBM_GUI_ACTIVE"] / n / 8          # shader cycles per launch (the counter sums the 8 XCDs)
simd = 1024
out = {"kernel": "k_pointnet_fwd", "launches": n, "note": "mean over the launches of tools/probes/pointnet_bench.py (512 / 1024 / 2048 / 8192 clouds x 400 points)",
       "cycles_per_launch": round(cyc), "mfma_busy_frac_per_simd": round(acc["SQ_VALU_MFMA_BUSY_CYCLES"] / n / simd / cyc, 3),
       "valu_active_frac_per_simd": round(acc["SQ_ACTIVE_INST_VALU"] * 4 / n / simd / cyc, 3),
       "vector_instructions_per_mfma": round((acc["SQ_INSTS_VALU"] - acc["SQ_INSTS_MFMA"]) / acc["SQ_INSTS_MFMA"], 2),
       "wave_parked_frac": round(acc["SQ_WAIT_ANY"] / acc["SQ_WAVE_CYCLES"], 3),
       "wave_issue_stall_frac": round(acc["SQ_WAIT_INST_ANY"] / acc["SQ_WAVE_CYCLES"], 3)}
import os
json.dump(out, open(os.environ.get("PN_SQ_OUT", "gpurun_out/r06_pointnet_sq_counters.json"), "w"), indent=1)
print(json.dumps(out))
PY
