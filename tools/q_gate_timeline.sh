#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MX_TAIL_GATE=1
rm -rf /tmp/kt_y
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_y -- python $R/tools/eq_sweep.py --toggle --steps 12 --no-profile --ticks ${TICKS:-256} ${OVERLAP---overlap-tail} 2>&1 | grep strips
python - $(find /tmp/kt_y -name "*kernel_trace.csv" | head -1) <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
prev=None
for r in rows:
    n=r['Kernel_Name']
    if 'spec_tiled' in n or 'k_mixer' in n or 'copyBuffer' in n or 'fill' in n:
        s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
        tag='EQ ' if 'spec_tiled' in n else ('MIX' if 'k_mixer' in n else 'cpy')
        if tag=='cpy' and e-s<50: continue
        print(f"{tag} start {s:10.1f} dur {e-s:8.1f}" + (f"  since prev EQ start {s-prev:8.1f}" if tag=='EQ ' and prev else ''))
        if tag=='EQ ': prev=s
PY
