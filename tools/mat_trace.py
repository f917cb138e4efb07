import csv, glob, sys, collections
f = glob.glob('gpurun_out/prof_mat/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    key = 'spec' if 'spec_tiled' in n else 'repair' if 'repair' in n else 'env_ticks' if 'env_ticks' in n else 'mixer' if 'mixer' in n else None
    if key: by[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in by.items():
    print(k, len(v), 'first8', [round(x, 3) for x in v[3:11]], 'last10', [round(x, 3) for x in v[-10:]])
