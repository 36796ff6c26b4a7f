#!/usr/bin/env python
"""HBM-side traffic per kernel symbol from two separate rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction from the guide: FETCH_SIZE tallies a 128-byte request of a
wide (16 B/lane) coalesced read as 64 bytes, so wide streaming reads are doubled ("x2"); both raw and x2 are
given because narrower accesses (the 4-byte gathers of k_loss / k_gather_normalize) are uncalibrated.
Infinity-Cache hits are included in these counters (they are fabric requests, not DRAM transactions).
Averages are over ALL launches of a kernel symbol (mixed layer shapes).
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.split('(')[0]
    name = re.sub(r'^void ', '', name)
    return name.replace('igi::', '')


def load(path):
    agg = collections.OrderedDict()
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        a = agg.setdefault(k, collections.defaultdict(float))
        a[r['Counter_Name']] += float(r['Counter_Value'])
        seen[k].add(r['Dispatch_Id'])
    for k in agg:
        agg[k]['_launches'] = len(seen[k])
    return agg


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    rows = []
    for k, f in fetch.items():
        if not ('gemm' in k or k.startswith('k_')):
            continue
        w = write.get(k, {})
        n = max(int(f['_launches']), 1)
        nw = max(int(w.get('_launches', 0)), 1)
        hit, miss = w.get('TCC_HIT_sum', 0.0), w.get('TCC_MISS_sum', 0.0)
        rows.append({
            "kernel": k, "launches": n,
            "fetch_MB_per_launch_raw": round(f['FETCH_SIZE'] * 1024 / n / 1e6, 2),
            "fetch_MB_per_launch_x2": round(2 * f['FETCH_SIZE'] * 1024 / n / 1e6, 2),
            "write_MB_per_launch": round(w.get('WRITE_SIZE', 0.0) * 1024 / nw / 1e6, 2),
            "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss > 0 else None,
        })
    rows.sort(key=lambda r: -(r["fetch_MB_per_launch_x2"] + r["write_MB_per_launch"]) * r["launches"])
    print(json.dumps({"note": __doc__.strip().split("\n\n")[1].replace("\n", " ") if False else
                      "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum, separate passes, same "
                      "bench.py command; FETCH_SIZE/WRITE_SIZE are KiB; gfx950 FETCH_SIZE under-reads wide coalesced "
                      "streams by 2x (MI355X_MICROARCH.md, HBM section): both raw and x2 given; averages are over ALL "
                      "launches of a kernel symbol (mixed layer shapes); Infinity-Cache hits are included",
                      "kernels": rows}, indent=1))


if __name__ == "__main__":
    main()
