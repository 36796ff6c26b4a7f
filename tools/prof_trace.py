"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid): python tools/prof_trace.py <trace.csv> [top_n]"""
import csv, collections, re, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.OrderedDict()
for r in rows:
    name=r['Kernel_Name']
    m=re.search(r'gemm_f32_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)>',name)
    m2=re.search(r'gemm_dma_kernel<(\d+), (\w+), (\w+)>',name)
    if m: short='gemm<%s,%s,%s,%s>'%(m.group(1),m.group(2),m.group(5)[0],m.group(6)[0])
    elif m2: short='dma<%s,%s,%s>'%(m2.group(1),m2.group(2)[0],m2.group(3)[0])
    else: short=name.split('(')[0][:40]
    key=(short, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),r['Grid_Size_Y'],r['Grid_Size_Z'])
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    a=agg.setdefault(key,[0,0]); a[0]+=1; a[1]+=d
tot=sum(a[1] for a in agg.values())
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
    print(f"{k[0]:30s} wgs=({k[1]},{k[2]},{k[3]}) n={a[0]:5d} avg={a[1]/a[0]/1e3:8.1f}us tot={a[1]/1e6:8.2f}ms {100*a[1]/tot:5.1f}%")
