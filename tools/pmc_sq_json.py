#!/usr/bin/env python
"""A rocprofv3 --pmc counter_collection.csv as JSON: per kernel (name up to its argument list), the mean of each counter over its dispatches
and the number of dispatches -- what bench.py derives `roofline.limiter`'s figures from.   python tools/pmc_sq_json.py <csv> <out.json> [family]"""
import collections
import csv
import json
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from kernel_hash import kernel_hash  # noqa: E402

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: dict({c: agg[k][c] / cnt[k][c] for c in agg[k]}, dispatches=max(cnt[k].values())) for k in agg}
json.dump({"kernel_sources_sha16": kernel_hash(sys.argv[3] if len(sys.argv) > 3 else "audio"), "mean_per_dispatch": out}, open(sys.argv[2], "w"), indent=1)
