#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_fc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_fc -- python $R/bench.py --force-combine --no-cpu-baseline --no-realtime --no-north-star --no-material-leg --no-rate-leg --no-contract-leg --fir-ticks 0 --repeats 0 --video-frames 0 --no-t-sweep --no-held-leg --no-headline-parity --steps 6 --warmup 2 > /dev/null 2>&1
python - $(find /tmp/kt_fc -name "*kernel_trace.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
eqs=[i for i,r in enumerate(rows) if 'spec_tiled' in r['Kernel_Name']]
lo=int(rows[eqs[-4]]['Start_Timestamp'])-200000; hi=int(rows[eqs[-2]]['End_Timestamp'])+200000
sel=[r for r in rows if lo<=int(r['Start_Timestamp'])<=hi]
for r in sel:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print(f"{r['Kernel_Name'].split('(')[0][-44:]:46s} q={r.get('Queue_Id','?'):>3s} start {s:10.1f} dur {e-s:8.1f}")
PY
