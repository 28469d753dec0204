#!/usr/bin/env python3
"""Timeline of one MD step of the calculator from a rocprofv3 kernel trace (csv): kernels in start
order with duration and the idle gap in front of each, averaged over the steady steps.
  python scripts/md_trace_summary.py <kernel_trace.csv> [first kernel name prefix]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fp:
    for r in csv.DictReader(fp):
        name = r["Kernel_Name"].replace("void ", "").replace("ta::(anonymous namespace)::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0]))
rows.sort()
first = sys.argv[2] if len(sys.argv) > 2 else "filter_kernel<0>"
starts = [k for k, r in enumerate(rows) if r[2].startswith(first)]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = steps[len(steps) // 2:]  # steady half
n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == n]
print(f"{len(steps)} steps of {n} kernels")
tot = 0.0
for k in range(n):
    dur = sum(s[k][1] - s[k][0] for s in steps) / len(steps) / 1e3
    gap = sum((s[k][0] - s[k - 1][1]) if k else 0 for s in steps) / len(steps) / 1e3
    off = sum(s[k][0] - s[0][0] for s in steps) / len(steps) / 1e3
    print(f"{off:8.1f} us  gap {gap:6.1f}  dur {dur:6.1f}  {steps[0][k][2][:60]}")
span = sum(s[-1][1] - s[0][0] for s in steps) / len(steps) / 1e3
period = sum(b[0][0] - a[0][0] for a, b in zip(steps[:-1], steps[1:])) / max(1, len(steps) - 1) / 1e3
print(f"first kernel start -> last kernel end: {span:.1f} us; step period {period:.1f} us")
