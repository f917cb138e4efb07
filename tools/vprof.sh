# per-kernel times of the video leg, fused and unfused launches (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in fused unfused; do
  if [ $mode = unfused ]; then export MX_VIDEO_NO_LAUNCH_FUSION=1; else unset MX_VIDEO_NO_LAUNCH_FUSION; fi
  rm -rf /tmp/vp_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp_$mode -- python $R/tools/vleg.py 1920 > $R/gpurun_out/vprof_$mode.log 2>&1
  f=$(find /tmp/vp_$mode -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/vprof_${mode}_kernel_stats.csv
  echo "== $mode"; head -8 "$f" | cut -c1-200; grep value $R/gpurun_out/vprof_$mode.log | tail -1
done
