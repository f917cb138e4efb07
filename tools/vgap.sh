cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/vg; rocprofv3 --kernel-trace --output-format csv -d /tmp/vg -- python $R/tools/vleg.py 3840 > /dev/null 2>&1
python - $(find /tmp/vg -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
vb = [r for r in rows if "k_video_batch" in r["Kernel_Name"]]
gaps = []; durs = []
for a, b in zip(vb, vb[1:]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3); durs.append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
import statistics
print("launches", len(vb), "median dur us", statistics.median(durs), "median gap us", statistics.median(gaps), "mean gap", sum(gaps) / len(gaps))
print("first 40 (dur, gap):", [(round(d), round(g, 1)) for d, g in list(zip(durs, gaps))[20:60]])
PY
