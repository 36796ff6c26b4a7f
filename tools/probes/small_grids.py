"""kernels of a rocprofv3 --kernel-trace CSV that run long on few workgroups: (name, workgroups, calls, avg us)"""
import csv, glob, sys, collections, re
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])) if "Grid_Size_X" in r else int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    name = re.sub(r"\(.*$", "", r["Kernel_Name"])[:70]
    a = acc[(name, wg)]
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = sorted(((n, wg, c, t / c) for (n, wg), (c, t) in acc.items()), key=lambda x: -x[3] * x[2])
for n, wg, c, us in rows:
    if c >= 32 and us >= 8 and wg < 512:
        print(f"{us:8.1f} us x{c:5d}  {wg:6d} WGs  {n}")
