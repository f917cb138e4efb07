#!/usr/bin/env python
"""The clock the chip sustains under each kernel: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / 8 / the duration of the same
dispatches (the pass's own kernel trace).   python tools/pmc_clock.py <counter_collection.csv> <kernel_trace.csv> <out.json>"""
import collections
import csv
import json
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from kernel_hash import kernel_hash  # noqa: E402

cc, kt, out = sys.argv[1:4]
busy = collections.defaultdict(float)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        busy[r["Dispatch_Id"]] += float(r["Counter_Value"])
dur, name = {}, {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); name[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for d, b in busy.items():
    if d in dur and dur[d] > 100000 and "mx::" in name[d] and "k_tail_gate" not in name[d]:   # (this library's kernels; not the runtime's copies, not the one-wave gate)         # the GRBM method is not usable below ~100 us (it reported 2.84-3.06 GHz, above the 2.4 GHz maximum, for the short kernels)
        a = agg[name[d]]; a[0] += b / 8.0; a[1] += dur[d]; a[2] += 1
res = {k: {"ghz": round(v[0] / v[1], 3), "dispatches": v[2]} for k, v in agg.items()}
json.dump({"kernel_sources_sha16": kernel_hash("audio"), "ghz_by_kernel": res,
           "how": "GRBM_GUI_ACTIVE / 8 XCDs / duration (ns) of the same dispatches, mean over dispatches longer than 100 us (shorter ones are not listed: the method does not resolve them)"}, open(out, "w"), indent=1)
print(json.dumps(res))
