#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_* counter_collection.csv: per kernel, counter totals / launches / ray-waves."""
import csv, json, sys
from collections import defaultdict
path, nwaves = sys.argv[1], float(sys.argv[2])
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name']
    if 'march_kernel' not in k and 'crossings_kernel' not in k:
        continue
    k = k.split('(')[0][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); launches[k].add(r['Dispatch_Id'])
    acc[k]['VGPR'] = float(r['VGPR_Count']); acc[k]['scratch'] = float(r['Scratch_Size'])
for k, d in acc.items():
    n = len(launches[k])
    print(k, n, 'launches:', json.dumps({c: (round(v / n / nwaves, 1) if c not in ('VGPR', 'scratch') else v) for c, v in sorted(d.items())}))
