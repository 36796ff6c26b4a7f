"""print the kernels of a rocprofv3 --stats CSV that are not this library's, with launches per optimizer step"""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
other = 0.0
for r in rows:
    n = r["Name"]
    if "igi::" in n:
        continue
    c = int(r["Calls"])
    if c / steps < 0.9:
        continue
    other += float(r["TotalDurationNs"])
    print(f"{c / steps:6.2f}/step {float(r['AverageNs']) / 1e3:7.1f} us  {n[:150]}")
print(f"non-library kernels: {other / steps / 1e3:.1f} us/step of {tot / steps / 1e3:.1f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    print(f"{int(r['Calls']) / steps:6.2f}/step {float(r['AverageNs']) / 1e3:7.1f} us {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
