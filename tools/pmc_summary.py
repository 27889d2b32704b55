#!/usr/bin/env python3
"""Average the PMC counters of the conv kernels in rocprofv3 counter_collection CSVs."""
import csv, glob, os, sys, collections
for d in sys.argv[1:]:
    files = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                      # a directory accumulates one file per run: the newest
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'conv_' not in k and 'wgrad_' not in k:
                continue
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, c in acc.items():
            print(d, k)
            for name, v in sorted(c.items()):
                print(f'   {name:28s} {sum(v)/len(v):16.0f}  (n={len(v)})')
