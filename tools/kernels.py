"""Pretty-print the per-kernel table of a bench.py JSON line read from stdin."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms")
for c in d.get("kernels", []):
    print("   %-40s n=%3d avg=%7.2f us  tot=%6.3f ms  %6.1f TF" % (c["name"], c["launches_per_update"], c["avg_us"],
                                                                    c["ms_per_update"], c["tflops"]))
