#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, the mean of each counter over its dispatches."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-60:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print(f"   {c:34s} {agg[k][c] / cnt[k][c]:18.1f}  (n={cnt[k][c]})")
