#!/bin/bash
# the video leg with the scaler tiles of batch k + 1 as their OWN launch on a second stream and the chain launch of batch k behind a gate
# (MX_VIDEO_SPLIT=1; 2 = two streams, no gate) against the one-launch default (0).  Interleaved repeats on one box; then the bit-exactness of the split form
# (the video graph tests under MX_VIDEO_SPLIT=1) and a kernel trace of it (do the two launches overlap?).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vgate; mkdir -p $O
for rep in 1 2 3; do
  for m in 0 1 2 3 4; do
    echo "split=$m $(MX_VIDEO_SPLIT=$m python $R/tools/vleg.py 3840 1 main 2>/dev/null | tail -1)"
  done
done | tee $O/times.txt
(cd $R && MX_VIDEO_SPLIT=1 python -m pytest tests/test_gpu_video_graph.py -q -m gpu -x 2>&1 | tail -3) | tee $O/parity_split.txt
rm -rf /tmp/vg_kt; MX_VIDEO_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/vg_kt -- python $R/tools/vleg.py 1920 1 main > /dev/null 2>&1
python - <<'PY' | tee $O/timeline.txt
import csv, glob
f = glob.glob('/tmp/vg_kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_video_batch<3' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# pair launches: a jobs launch (side stream) and the chains launch that follows
by_stream = {}
for r in rows:
    by_stream.setdefault(r.get('Stream_Id', r.get('Queue_Id')), []).append(r)
print({k: len(v) for k, v in by_stream.items()})
ks = sorted(by_stream, key=lambda k: -len(by_stream[k]))[:2]
if len(ks) == 2:
    a, b = by_stream[ks[0]], by_stream[ks[1]]
    tot = ov = 0
    j = 0
    for ra in a[20:120]:
        s0, e0 = int(ra['Start_Timestamp']), int(ra['End_Timestamp'])
        for rb in b:
            s1, e1 = int(rb['Start_Timestamp']), int(rb['End_Timestamp'])
            if e1 < s0 or s1 > e0: continue
            ov += min(e0, e1) - max(s0, s1)
        tot += e0 - s0
    da = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in a[20:120]) / 100
    db = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in b[20:120]) / max(1, len(b[20:120]))
    print(f"stream A avg {da/1e3:.1f} us, stream B avg {db/1e3:.1f} us, share of A's time overlapped by B: {ov/max(1,tot):.3f}")
PY
