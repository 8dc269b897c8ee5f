#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of each kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv"))):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(d)):
        key = (r["Kernel_Name"][:60], r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1] += 1
    print("#", d)
    for (k, c), (s, n) in sorted(acc.items()):
        print(f"{k:60s} {c:28s} n={n:4d} mean={s / n:.6g}")
