#!/usr/bin/env python3
"""Print a rocprofv3 --kernel-trace --stats kernel_stats.csv per train step: python tools/kstats.py <csv> <steps> [top]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms/step", round(tot / 1e6 / steps, 3), "launches/step", round(sum(int(r["Calls"]) for r in rows) / steps, 1))
for r in rows[:top]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']) / steps:5.1f} {float(r['TotalDurationNs']) / 1e3 / steps:8.1f} us/step {float(r['AverageNs']) / 1e3:7.1f} us")
