#!/bin/bash
# The colour-matrix experiment (VERDICT r4 task 3a): the video leg with the matrix as packed f32 FMAs (default) and on the matrix cores (MX_VIDEO_MFMA_MATRIX=1):
# device time per frame over several repetitions, then one SQ counter pass each.  gpurun -- 'bash tools/vmfma.sh r05'
set -u
R=${1:-r05}; REPO=$(pwd); OUT=$REPO/gpurun_out/$R; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
hipcc --offload-arch=gfx950 -O3 $REPO/tools/mfma_probe.hip -o /tmp/mfma_probe 2>/dev/null && /tmp/mfma_probe > $OUT/mfma_probe.txt 2>&1; cat $OUT/mfma_probe.txt
for m in 0 1; do
  echo "== MX_VIDEO_MFMA_MATRIX=$m"; MX_VIDEO_MFMA_MATRIX=$m python $REPO/tools/vleg.py 1920 4 main | tee $OUT/vmfma_times_$m.txt
done
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
for m in 0 1; do
  rm -rf /tmp/p_vm$m
  MX_VIDEO_MFMA_MATRIX=$m timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/p_vm$m -- python $REPO/tools/vleg.py 1280 1 main > /dev/null 2>&1
  f=$(find /tmp/p_vm$m -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summary.py $f > $OUT/video_sq_mfma_$m.txt
  grep -A9 "k_video_batch" $OUT/video_sq_mfma_$m.txt | head -24
done
