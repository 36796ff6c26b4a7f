"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: python tools/prof_pmc.py <counter_collection.csv>"""
import csv, collections, re, sys
def load(path):
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name=r['Kernel_Name']
        m=re.search(r'gemm_f32_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)>',name)
        short = ('gemm<%s,%s,%s,%s>'%(m.group(1),m.group(2),m.group(5)[0],m.group(6)[0])) if m else name.split('(')[0][:28]
        key=(short, r['Grid_Size'])
        a=agg.setdefault(key,collections.defaultdict(float))
        a[r['Counter_Name']]+=float(r['Counter_Value']); a['_n_'+r['Counter_Name']]+=1
        a['_t']+= (int(r['End_Timestamp'])-int(r['Start_Timestamp']))
        a['_nt']+=1
    return agg
if __name__=='__main__':
    a1=load(sys.argv[1])
    for k,v in a1.items():
        if 'igi' not in k[0] and 'gemm' not in k[0]: continue
        names=[c for c in v if not c.startswith('_')]
        n=v['_n_'+names[0]]
        s=f"{k[0]:24s} grid={k[1]:>8s} n={int(n):4d} t={v['_t']/v['_nt']/1e3:7.1f}us "
        for c in names: s+=f"{c}={v[c]/n:.3g} "
        print(s)
