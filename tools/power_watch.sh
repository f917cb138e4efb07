# the board's power and shader clock while a command runs (rocm-smi sampled in a loop).  gpurun -- 'bash tools/power_watch.sh python bench.py ...'
cd $GRAFT_REPO_ROOT
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/pw.log 2>&1 &
W=$!
"$@" > /tmp/cmd.out 2> /tmp/cmd.err
kill $W
tail -c 400 /tmp/cmd.out | head -c 400; echo
python - <<'PY'
import re
p=[]; c=[]
for ln in open('/tmp/pw.log'):
    m=re.search(r'Power \(W\): ([0-9.]+)', ln) or re.search(r'Power.*?: ([0-9.]+)', ln)
    k=re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', ln)
    if m and k: p.append(float(m.group(1))); c.append(int(k.group(1)))
print("samples", len(p))
if p:
    import statistics
    print("power W: max", max(p), "median", statistics.median(p)); print("sclk MHz: max", max(c), "min", min(c))
    busy=[(a,b) for a,b in zip(p,c) if a > 0.5*max(p)]
    print("under load (power > half its max):", len(busy), "samples, median power", statistics.median([a for a,_ in busy]), "W, median sclk", statistics.median([b for _,b in busy]), "MHz")
PY
head -3 /tmp/pw.log
