#!/bin/bash
# every dispatch of two consecutive steps of the headline job, in order: name, start (us after the first), duration (us), queue.   gpurun -- 'bash tools/step_timeline.sh'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/st_kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/st_kt -- python $R/bench.py --headline-only --no-headline-parity --steps 10 --warmup 2 --full-out /tmp/st_full.json > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/st_kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
eq = [i for i, r in enumerate(rows) if 'k_eq_three_spec_tiled' in r['Kernel_Name']]
a, b = eq[6], eq[8]
t0 = int(rows[a - 3]['Start_Timestamp'])
prev_end = {}
for r in rows[a - 3:b + 1]:
    q = r.get('Queue_Id', '?')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print(f"{r['Kernel_Name'].split('(')[0][-52:]:52s} q{q} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap on its queue {gap:7.1f} us  grid {r.get('Grid_Size_X', '')} wg {r.get('Workgroup_Size_X', '')}")
PY
