cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/mp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/mp -- python $R/bench.py --no-cpu-baseline --no-t-sweep --no-realtime --no-north-star --fir-ticks 0 --video-frames 0 --repeats 0 --steps 6 --no-held-leg > /dev/null 2>&1
f=$(find /tmp/mp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# per-kernel durations in dispatch order: print the sequence of (name, us) for the repair kernel over time
names = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-40:]
    names[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in names.items():
    if "eq_three" in n or "env_ticks" in n or "mixer" in n:
        print(n, len(v), "first10", [round(x) for x in v[:10]], "last12", [round(x) for x in v[-12:]])
PY
