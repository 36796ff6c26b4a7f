# SQ issue-side counters of the teacher update's kernels (one rocprofv3 --pmc pass over bench.py, teacher only):
# per (kernel, grid): matrix-busy and vector-active cycles per SIMD as fractions of the launch, vector instructions per
# MFMA, parked / issue-stalled wave cycles -> gpurun_out/r06_teacher_sq_counters.json
# (launch durations under --pmc are longer than in the timed bench: the counters serialise dispatches)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/t_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
  --kernel-include-regex "gemm_dma|k_env_fwd|k_fwd12|k_trunk_loss|k_loss|k_latent_bwd|k_slab_reduce|k_adam_gather|k_sumsq|k_rb_level" -d gpurun_out/t_sq -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-student --no-experiments --no-peak-probe > gpurun_out/t_sq.out 2>&1
python3 - <<'PY'
import csv, glob, collections, json, re
f = glob.glob("gpurun_out/t_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    name = re.sub(r"^void |igi::|\(.*$", "", r["Kernel_Name"]).strip()
    key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[key] += r["Counter_Name"] == "SQ_WAVE_CYCLES"
rows = []
for k in sorted(acc, key=lambda k: -acc[k]["GRBM_GUI_ACTIVE"]):
    a, c = acc[k], n[k]
    if c < 32: continue          # the per-step kernels only
    cyc = a["GRBM_GUI_ACTIVE"] / c / 8
    mf = a["SQ_INSTS_MFMA"]
    rows.append({"kernel": k[0], "workgroups": k[1], "launches": c, "cycles_per_launch_under_pmc": round(cyc),
                 "mfma_busy_frac_per_simd": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / c / 1024 / cyc, 3),
                 "valu_active_frac_per_simd": round(a["SQ_ACTIVE_INST_VALU"] * 4 / c / 1024 / cyc, 3),
                 "vector_instructions_per_mfma": round((a["SQ_INSTS_VALU"] - mf) / mf, 2) if mf else None,
                 "wave_parked_frac": round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 3),
                 "wave_issue_stall_frac": round(a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 3)})
json.dump({"source": "rocprofv3 --pmc (SQ_*, GRBM_GUI_ACTIVE) over bench.py --steps 2 --no-student", "kernels": rows},
          open("gpurun_out/r06_teacher_sq_counters.json", "w"), indent=1)
for r in rows[:14]: print(r)
PY
