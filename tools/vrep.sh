# video leg at several K (ticks per launch): device us per frame (hipEvents) and the rocprof kernel time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in 1 2 4 8 12 16; do
  export MX_VIDEO_BATCH=$k
  echo "K=$k $(python $R/tools/vleg.py 1920 2>/dev/null | tail -1)"
done
export MX_VIDEO_BATCH=8
rm -rf /tmp/vr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vr -- python $R/tools/vleg.py 1920 > /dev/null 2>&1
head -6 $(find /tmp/vr -name "*kernel_stats.csv" | head -1) | cut -c1-160
