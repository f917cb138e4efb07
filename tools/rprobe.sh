# How long the proof / repair pass of the speculative EqThree takes on ONE strip of a given material, under rocprofv3 --kernel-trace:
# the planner's own plan for a lone strip (2048 one-tick chunks) and the plan of the 1024-strip bench (256 chunks of 6400 samples).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for chunks in 0 256; do
for k in noise muted gaps onegap; do
rm -rf /tmp/rp
echo "== MX_EQ_SPEC_CHUNKS=$chunks $k"
MX_EQ_SPEC_CHUNKS=$chunks rocprofv3 --kernel-trace --output-format csv -d /tmp/rp -- python $R/tools/repair_probe.py $k 2>/dev/null | grep chunks_run
f=$(find /tmp/rp -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for key in ("repair", "spec"):
    v = [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if key in r["Kernel_Name"]]
    print("   ", key, "us:", v)
PY
done
done
