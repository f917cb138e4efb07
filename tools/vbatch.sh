#!/bin/bash
# VERDICT r4 task 3c: does a SMALLER ring of scaled frames (K frames per launch, a Scaler writes 2K output frames in turn) keep the scaled layers in the 256 MiB
# Infinity Cache between the scaler tiles of one launch and the chains of the next?  Device time per frame and FETCH_SIZE / WRITE_SIZE per frame at K = 4, 8, 16.
# gpurun -- 'bash tools/vbatch.sh r05'
set -u
R=${1:-r05}; REPO=$(pwd); OUT=$REPO/gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
: > $OUT/vbatch.txt
for k in 4 8 16; do
  echo "== MX_VIDEO_BATCH=$k" | tee -a $OUT/vbatch.txt
  MX_VIDEO_BATCH=$k python $REPO/tools/vleg.py 1920 3 main | tee -a $OUT/vbatch.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_vb; MX_VIDEO_BATCH=$k timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_vb -- python $REPO/tools/vleg.py 1280 1 main > /dev/null 2>&1
    f=$(find /tmp/p_vb -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" $c $k <<'PY' | tee -a $OUT/vbatch.txt
import csv, sys
f, c, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = n = 0
for r in csv.DictReader(open(f)):
    if "k_video_batch" in r["Kernel_Name"] and r["Counter_Name"] == c:
        tot += float(r["Counter_Value"]); n += 1
# FETCH_SIZE / WRITE_SIZE are reported in KiB-like units by this rocprofv3 (x 1024 -> bytes is what tools/pmc_traffic.py applies); print raw and per dispatch
print(f"{c}: {n} dispatches of k_video_batch, raw sum {tot:.4g}, per dispatch {tot / max(1, n):.4g} (K = {k} frames per dispatch -> per frame {tot / max(1, n) / k:.4g})")
PY
  done
done
