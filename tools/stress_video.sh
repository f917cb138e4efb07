#!/bin/bash
# the video-side differential stress tools (cascade, VideoMixer, scaler, row bands, monitor sink, coverage planes).   gpurun -- 'bash tools/stress_video.sh [first seed]'
cd $GRAFT_REPO_ROOT
F=${1:-50000}
for t in "stress_cascade.py $F 60" "stress_vmixer.py $F 40" "stress_scaler.py $F 40" "stress_bands.py $F 30" "stress_monitor.py $F 20" "stress_alpha.py $F 120"; do
  echo "== $t"; timeout 1200 python tools/$t 2>&1 | tail -2
done
