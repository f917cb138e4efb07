# SQ / memory counters of the video leg's kernels (fused launch and the two kernels alone); rocprofv3 --pmc, kernel trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-vpmc}
mkdir -p $OUT
for mode in fused; do
  if [ $mode = unfused ]; then export MX_VIDEO_NO_LAUNCH_FUSION=1; else unset MX_VIDEO_NO_LAUNCH_FUSION; fi
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf /tmp/vq
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/vq -- python $R/tools/vleg.py 640 > /dev/null 2>&1
    f=$(find /tmp/vq -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python $R/tools/pmc_summary.py $f > $OUT/${mode}_set$i.txt
  done
done
grep -h -A9 "k_video_batch<2>\|k_fade_chain_rgba\|k_scale_bicubic_tiled" $OUT/*.txt | head -150
