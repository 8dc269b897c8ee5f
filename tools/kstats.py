#!/usr/bin/env python3
"""Compact view of a rocprofv3 kernel_stats.csv: python tools/kstats.py FILE [N]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:n]:
    name = re.sub(r"ucg::\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    print(f"{name[:64]:64s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:10.1f} us  {float(r['Percentage']):6.2f} %")
