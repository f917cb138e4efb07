cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/np; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -- python $R/tools/ctl_probe.py 2048 > /dev/null 2>&1
python - $(find /tmp/np -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'eq_three' in r['Name'] or 'mixer' in r['Name'] or 'envelope' in r['Name']:
        print(r['Name'][:90], r['Calls'], round(float(r['AverageNs'])/1e6,3))
PY
