#!/usr/bin/env python3
'''Per-kernel average of every counter in a rocprofv3 --pmc csv output directory.'''
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, ctrs in acc.items():
    n = max(len(v) for v in ctrs.values())
    print(f'{k}  (dispatches: {n})')
    for c, v in sorted(ctrs.items()):
        print(f'    {c:32s} avg {sum(v) / len(v):16.1f}')
