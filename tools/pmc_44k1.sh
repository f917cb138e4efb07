#!/bin/bash
# FETCH_SIZE / WRITE_SIZE and the SQ counters of the headline job at 44.1 kHz (the RT instantiations of the tiled EqThree kernel).  gpurun -- 'bash tools/pmc_44k1.sh'
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
COMMON="--headline-only --steps 4 --warmup 1 --sample-rate 44100 --full-out /tmp/bench_full_44k1.json"
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  rm -rf /tmp/p44; timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p44 -- python $REPO/bench.py $COMMON > /dev/null 2>&1
  python $REPO/tools/pmc_summary.py $(find /tmp/p44 -name "*counter_collection.csv" | head -1) | grep -A6 "spec_tiled"
done
