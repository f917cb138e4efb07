# kernel trace of T = 64 submissions (1024 strips): per-kernel durations and the gaps between them.  gpurun -- 'bash tools/t64.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FL="--ticks-per-step 64 --steps 40 --warmup 5 --no-cpu-baseline --no-realtime --no-t-sweep --no-north-star --no-held-leg --no-material-leg --no-scaling-probe --fir-ticks 0 --repeats 0 --video-frames 0 $T64_FLAGS"
python $R/bench.py $FL | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', l['value'], 'ms/step', l['ms_per_step'], l['roofline'].get('kernel_ms_per_step'))"
rm -rf /tmp/t64; rocprofv3 --kernel-trace --output-format csv -d /tmp/t64 -- python $R/bench.py $FL > /dev/null 2>&1
python - $(find /tmp/t64 -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # the timed steps
seq = [(r["Kernel_Name"].split("(")[0][-40:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# one period of the steady state: from one k_env_ticks to the next
idx = [i for i, s in enumerate(seq) if "k_env_ticks" in s[0]]
if len(idx) > 3:
    a, b = idx[-3], idx[-2]
    t0 = seq[a][1]
    for n, s, e in seq[a:b + 1]:
        print(f"{n:42s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in seq: agg[n][0] += e - s; agg[n][1] += 1
for n, (t, c) in agg.items(): print(f"{n:42s} n {c:4d} mean {t / c / 1e3:7.1f} us")
PY
