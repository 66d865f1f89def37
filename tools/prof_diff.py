#!/usr/bin/env python3
"""per-step kernel time of two rocprofv3 kernel-stats CSVs side by side: python tools/prof_diff.py A.csv B.csv"""
import csv, sys
def load(f):
    rows = list(csv.DictReader(open(f)))
    steps = max(int(r["Calls"]) for r in rows if "adamw_tick" in r["Name"])
    d = {}
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d[n] = d.get(n, 0.0) + float(r["TotalDurationNs"]) / steps / 1e3
    return d, steps
a, sa = load(sys.argv[1]); b, sb = load(sys.argv[2])
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, 0) - b.get(k, 0)))
print("steps", sa, sb, "total us/step", round(sum(a.values()), 1), round(sum(b.values()), 1))
for k in keys[:25]:
    print("%-70s %8.1f %8.1f %+8.1f" % (k[:70], a.get(k, 0), b.get(k, 0), b.get(k, 0) - a.get(k, 0)))
