#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ht
timeout 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/ht -- python $R/tools/eq_sweep.py --toggle --steps 40 --no-profile --ticks 256 --overlap-tail 2>&1 | grep strips
f=$(find /tmp/ht -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-150
t=$(find /tmp/ht -name "*hip_api_trace.csv" | head -1)
python - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
big=[r for r in rows if int(r['End_Timestamp'])-int(r['Start_Timestamp'])>2_000_000]
for r in big[-12:]:
    print(r['Function'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us')
PY
