#!/usr/bin/env python3
"""Launches of ONE steady-state train step from a rocprofv3 kernel trace (the step starts at mul_keep_kernel, train.py:181):
per-kernel counts and time, and whether any ATen / runtime kernel is left on the step.  kernel_stats.csv divided by the step
count also spreads the start-up launches (initialisation fills, first weight preparation) over the steps.
    python tools/step_launches.py <..._kernel_trace.csv>"""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("mul_keep_kernel")]
gaps = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
print(f"{len(marks)} steps in the trace; launches between consecutive step starts, last six: {gaps[-6:]}")
step = rows[marks[-2]:marks[-1]]
cnt, dur = collections.Counter(), collections.Counter()
for r in step:
    k = r["Kernel_Name"].split("(")[0]
    cnt[k] += 1
    dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"steady-state step: {len(step)} launches, {sum(dur.values()) / 1e6:.3f} ms of kernel time")
foreign = {k: cnt[k] for k in cnt if "at::" in k or "rocclr" in k}
print("ATen / runtime kernels on the step:", foreign if foreign else "none")
for k, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{n:4d} {dur[k] / 1e3:9.1f} us  {k[:110]}")
