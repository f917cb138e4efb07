#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, MI355X_MICROARCH.md) into HBM bytes per launch
per kernel and per bench launch group:   python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <out.md>
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section), so it is doubled."""
import collections
import csv
import json
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from kernel_hash import kernel_hash  # noqa: E402

GROUPS = {   # bench.py's launch groups (what its hipEvents bracket)
    "eq_three": ("k_env_ticks", "k_eq_three_spec", "k_eq_three_repair", "k_eq_three_scan", "k_eq_three_wave", "k_eq_three_exact"),
    "mixer": ("k_mixer",),
    "video_scaler": ("k_scale_bicubic",),
    "video_chain": ("k_fade_chain",),
    "video_batch": ("k_video_batch<",),
    "fir": ("k_fir(", "k_fir<"),
    "resample": ("k_resample(", "k_resample<", "k_resample_ps<"),
}
FC_SPLIT = ("fir", "resample")


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    return {k: (tot[k] / max(1, len(disp[k])), len(disp[k])) for k in tot}


def main():
    fetch, write, out_json, out_md = sys.argv[1:5]
    config = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
    family = sys.argv[6] if len(sys.argv) > 6 else "audio"
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w)):
        fb = f.get(k, (0.0, 0))[0] * 1024.0 * 2.0
        wb = w.get(k, (0.0, 0))[0] * 1024.0
        rows.append((k, fb, wb, max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1])))
    groups = {}
    for g, names in GROUPS.items():
        # a group's launch = one dispatch of each of its kernels in a step: sum of the per-dispatch means of the LONG-stream variants
        sel = [r for r in rows if any(n in r[0] for n in names)]
        # the FIR leg runs its chain in both arithmetic orders in one process: <..., true> instantiations are the contracted launches
        for suffix, part in (("", [r for r in sel if not (g in FC_SPLIT and ", true>(" in r[0])]), ("_fc", [r for r in sel if g in FC_SPLIT and ", true>(" in r[0]])):
            if part:
                groups[g + suffix] = sum(r[1] + r[2] for r in part if r[1] + r[2] > 0.01 * max(x[1] + x[2] for x in part))
    json.dump({"config": config, "kernel_sources_sha16": kernel_hash(family), "bytes_per_launch": groups,
               "per_kernel": {r[0][:120]: {"fetch_bytes_x2": r[1], "write_bytes": r[2], "dispatches": r[3]} for r in rows}}, open(out_json, "w"), indent=1)
    with open(out_md, "w") as fh:
        fh.write("| kernel | dispatches | FETCH_SIZE x 2 (B / dispatch) | WRITE_SIZE (B / dispatch) | sum |\n|---|---|---|---|---|\n")
        for k, fb, wb, n in rows:
            fh.write(f"| `{k[:110]}` | {n} | {fb:.4g} | {wb:.4g} | {fb + wb:.4g} |\n")
        fh.write("\nlaunch groups (what bench.py's hipEvents bracket): " + json.dumps({k: round(v) for k, v in groups.items()}) + "\n")


if __name__ == "__main__":
    main()
