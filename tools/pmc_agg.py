#!/usr/bin/env python
"""Mean PMC counter values per (kernel symbol, grid) over all launches, from rocprofv3 counter_collection.csv files:
    python tools/pmc_agg.py <counter_collection.csv> [...]
"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r'^void ', '', r['Kernel_Name'].split('(')[0]).replace('igi::', '')[:44]
        k = (name, r.get('Grid_Size', ''))
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k in sorted(agg, key=lambda k: -agg[k].get('SQ_BUSY_CU_CYCLES', agg[k].get('SQ_WAVE_CYCLES', 0.0))):
    vals = "  ".join("%s=%.4g" % (c, v / max(len(seen[(k, c)]), 1)) for c, v in sorted(agg[k].items()))
    print("%-44s grid %-9s %s" % (k[0], k[1], vals))
