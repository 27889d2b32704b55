#!/usr/bin/env python3
"""Durations of selected kernels inside ONE steady-state iteration of a tools/kstat.sh trace (launch index within the
iteration, name fragment, microseconds).  usage: tools/iter_rows.py <kstat dir> <name fragment> [...]"""
import csv, glob, sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'onehot_rep' in n]
a, b = idx[-3], idx[-2]
tot = 0.0
for i in range(a, b):
    d = (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
    tot += d
    for frag in sys.argv[2:]:
        if frag in names[i]:
            print(f'{i - a:4d}  {frag:28s} {d:8.1f} us   grid {rows[i].get("Grid_Size_X", "")} x {rows[i].get("Grid_Size_Y", "")}')
gap = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3
print(f'{b - a} launches, kernel time {tot:.1f} us, wall {gap:.1f} us')
