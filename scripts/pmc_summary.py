#!/usr/bin/env python3
"""Median counter value per kernel-name substring from a rocprofv3 counter_collection.csv."""
import collections
import csv
import sys

path, needle = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
with open(path) as f:
    rd = csv.reader(f)
    head = next(rd)
    kn, cn, cv = head.index("Kernel_Name"), head.index("Counter_Name"), head.index("Counter_Value")
    for r in rd:
        if needle in r[kn]:
            acc[r[cn]].append(float(r[cv]))
for k, v in sorted(acc.items()):
    v.sort()
    print(f"{k:40s} n={len(v):4d} median={v[len(v) // 2]:.4g} max={v[-1]:.4g}")
