import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for c in d["kernels"]:
    if c["name"] in ("k_loss","other","k_slab_reduce","k_gather_stats","k_sumsq_stats"): print("   ", c["name"], c["avg_us"])
