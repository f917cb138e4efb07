cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in -1 16 32 64 128; do
rm -rf /tmp/np; PROBE_KINDS=envgate MX_ENV_SEGMENTS=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -- python $R/tools/ctl_probe.py 2048 > /dev/null 2>&1
python - $(find /tmp/np -name "*kernel_stats.csv" | head -1) $S <<'PY'
import csv,sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_env' in r['Name']:
        out.append((r['Name'].split('(')[0][-28:], r['Calls'], round(float(r['AverageNs'])/1e6,3)))
print('S', sys.argv[2], out)
PY
done
