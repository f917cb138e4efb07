#!/bin/bash
# Collect what profiles/rNN/ holds, on the GPU box:  gpurun -- 'bash tools/profile_round.sh r02'
# Every rocprofv3 pass is wrapped in its own `timeout`; PMC passes use --kernel-trace only (never the hip/hsa trace domains).
set -u
R=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
COMMON="--no-cpu-baseline --no-realtime --no-t-sweep --no-north-star --no-held-leg --fir-ticks 0 --repeats 0 --steps 4 --warmup 1 --video-frames 320"
# 1. the default command, as the driver runs it
timeout 900 python $REPO/bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err
# 2. the same command under the kernel trace
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/bench.py > $OUT/bench_line_under_rocprof.json 2>/dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_default_bench.csv
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md), headline configuration, fewer steps
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python $REPO/bench.py $COMMON > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python $REPO/bench.py $COMMON > /dev/null 2>&1
python $REPO/tools/pmc_traffic.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) \
    $OUT/pmc_traffic.json $OUT/pmc_hbm_traffic.md '{"strips": 1024, "ticks_per_step": 2048, "sample_rate": 48000, "fused": true, "eq_fast": false, "n_gpus": 1, "gates_toggle": true}'
# 4. SQ counters of the headline kernels, gates toggling and held
for mode in "" "--hold-gates"; do
  tag=toggle; [ -n "$mode" ] && tag=held
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/sq$tag -- python $REPO/bench.py $COMMON $mode > /dev/null 2>&1
  python $REPO/tools/pmc_summary.py $(find /tmp/sq$tag -name "*counter_collection.csv" | head -1) > $OUT/pmc_sq_$tag.txt
done
# 5. the clock the chip sustains under the headline kernel: busy cycles per SE / duration (GRBM_GUI_ACTIVE: per XCD)
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/clk -- python $REPO/bench.py $COMMON > /dev/null 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/clk -name "*counter_collection.csv" | head -1) > $OUT/pmc_clock.txt
cp $(find /tmp/clk -name "*kernel_trace.csv" | head -1) $OUT/pmc_clock_kernel_trace.csv 2>/dev/null
ls -la $OUT
