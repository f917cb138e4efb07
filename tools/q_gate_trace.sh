#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in plain gate; do
  rm -rf /tmp/kt_$mode
  if [ $mode = gate ]; then export MX_TAIL_GATE=1; else unset MX_TAIL_GATE; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$mode -- python $R/tools/eq_sweep.py --toggle --steps 6 --no-profile --ticks 256 --overlap-tail > /dev/null 2>&1
  f=$(find /tmp/kt_$mode -name "*kernel_trace.csv" | head -1)
  python - "$f" $mode <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
sel=[r for r in rows if any(k in r['Kernel_Name'] for k in ('spec_tiled','k_mixer','tail_gate','repair','env_ticks'))]
print(sys.argv[2])
for r in sel[-24:]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print(f"  {r['Kernel_Name'].split('(')[0][-40:]:42s} q={r.get('Queue_Id','?'):>3s} start {s:10.1f} us  dur {e-s:8.1f} us")
PY
done
