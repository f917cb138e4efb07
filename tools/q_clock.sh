# the clock under k_eq_three_spec_tiled for two shapes of the same per-SIMD work (one wave per SIMD, chunks of 4 ticks): several waves per strip / one wave per strip
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
FLB="--steps 12 --warmup 3 --no-cpu-baseline --no-realtime --no-t-sweep --no-north-star --no-held-leg --no-material-leg --no-rate-leg --no-contract-leg --no-scaling-probe --fir-ticks 0 --repeats 0 --video-frames 0 --no-headline-parity"
run() { # tag chunks strips ticks
  rm -rf /tmp/ck_$1
  MX_EQ_SPEC_CHUNKS=$2 timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/ck_$1 -- python $R/bench.py --strips $3 --ticks-per-step $4 $FLB > /dev/null 2>&1
  echo "$1: $(python $R/tools/pmc_clock.py $(find /tmp/ck_$1 -name '*counter_collection.csv' | head -1) $(find /tmp/ck_$1 -name '*kernel_trace.csv' | head -1) /tmp/ck_$1.json)"
}
run a_128x2048_c512 512 128 2048
run b_1024x256_c64 64 1024 256
run c_512x512_c128 128 512 512
