cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in noise gaps; do
rm -rf /tmp/rp
MX_EQ_SPEC_WARM=16 MX_EQ_SPEC_CHUNKS=256 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -- python $R/tools/repair_probe.py $k 2>/dev/null | grep chunks_run
f=$(find /tmp/rp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for key in ("repair", "spec"):
    v = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if key in r["Kernel_Name"]]
    print("   ", key, "us:", v)
PY
done
