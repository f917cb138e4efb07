#!/bin/bash
# The video leg with the chains and the scaler tiles as SEPARATE launches (MX_VIDEO_NO_LAUNCH_FUSION=1): each kernel's own time and SQ counters.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vsplit; mkdir -p $O
export MX_VIDEO_NO_LAUNCH_FUSION=1
rm -rf /tmp/vs_kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs_kt -- python $R/tools/vleg.py 1920 1 main > $O/line.json 2>/dev/null
cp $(find /tmp/vs_kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
pm() { tag=$1; shift; rm -rf /tmp/vs_$tag; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/vs_$tag -- python $R/tools/vleg.py 640 1 main > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/vs_$tag -name "*counter_collection.csv" | head -1) > $O/$tag.txt; }
pm sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pm sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
pm clk GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
head -8 $O/kernel_stats.csv | cut -c1-160; cat $O/line.json; grep -A9 "scale_bicubic_tiled\|fade_chain_rgba" $O/sq1.txt $O/sq2.txt $O/clk.txt
