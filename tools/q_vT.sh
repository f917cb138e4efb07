#!/bin/bash
# the video leg at several submission lengths (ticks per mx_graph_run_ticks call): what the pipeline's fill (one scale-only launch) and drain (one chains-only launch) cost per frame
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vgate; mkdir -p $O
for rep in 1 2; do for t in 256 512 1024 2048; do echo "T=$t $(VLEG_T=$t python $R/tools/vleg.py 8192 1 main 2>/dev/null | tail -1)"; done; done | tee $O/times_T.txt
rm -rf /tmp/vt_kt; VLEG_T=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vt_kt -- python $R/tools/vleg.py 8192 1 main > /dev/null 2>&1
head -6 $(find /tmp/vt_kt -name "*kernel_stats.csv" | head -1) | cut -c1-200 | tee $O/kernel_stats_T1024.csv
