#!/bin/bash
# Collect what profiles/rNN/ holds, on the GPU box:  gpurun -- 'bash tools/profile_round.sh r04'
# Every rocprofv3 pass is wrapped in its own `timeout`; PMC passes use --kernel-trace only (never the hip/hsa trace domains),
# FETCH_SIZE and WRITE_SIZE in separate runs (MI355X_MICROARCH.md).  profiles/rNN/README.md is generated from these outputs
# (tools/profile_readme.py), not written by hand.
set -u
R=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
COMMON="--headline-only --steps 4 --warmup 1 --full-out /tmp/bench_full_pmc.json"
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
pmc() {   # pmc <tag> <counters...> -- <command...>: one counter pass, summary into $OUT/<tag>.txt, raw csv path echoed
  local tag=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf /tmp/p_$tag
  timeout 900 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d /tmp/p_$tag -- "$@" > /dev/null 2>&1
  local f=$(find /tmp/p_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summary.py $f > $OUT/$tag.txt
  echo $f
}
# 2. the same command under the kernel trace
rm -rf /tmp/kt; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/bench.py --full-out /tmp/bench_full_kt.json > $OUT/bench_line_under_rocprof.json 2>/dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_default_bench.csv
# 2b. the headline's window as the kernel trace sees it (no counters): per-dispatch start / end in the default schedule and on one stream -> window_timeline.md
HL="--headline-only --no-headline-parity --steps 10 --warmup 2 --full-out /tmp/bench_full_hl.json"
rm -rf /tmp/kw_a /tmp/kw_1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kw_a -- python $REPO/bench.py $HL > /dev/null 2>&1
cp $(find /tmp/kw_a -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_headline_only.csv   # the headline's shape alone: the average the line's roofline.avg_launch_ms must agree with
MX_OVERLAP_AUTO=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kw_1 -- python $REPO/bench.py $HL > /dev/null 2>&1
python $REPO/tools/window_timeline.py $(find /tmp/kw_a -name "*kernel_trace.csv" | head -1) $(find /tmp/kw_1 -name "*kernel_trace.csv" | head -1) 1024 2048 > $OUT/window_timeline.md
# 3. HBM traffic of the headline configuration
FCSV=$(pmc pmc_fetch FETCH_SIZE -- python $REPO/bench.py $COMMON)
WCSV=$(pmc pmc_write WRITE_SIZE -- python $REPO/bench.py $COMMON)
python $REPO/tools/pmc_traffic.py $FCSV $WCSV $OUT/pmc_traffic.json $OUT/pmc_hbm_traffic.md \
    '{"strips": 1024, "ticks_per_step": 2048, "sample_rate": 48000, "fused": true, "eq_fast": false, "n_gpus": 1, "gates_toggle": true}' audio
# 3b. the same in the contracted order (MX_FLAG_FP_CONTRACT as the headline: --fp-contract)
FCSV=$(pmc pmc_fetch_fc FETCH_SIZE -- python $REPO/bench.py $COMMON --fp-contract)
WCSV=$(pmc pmc_write_fc WRITE_SIZE -- python $REPO/bench.py $COMMON --fp-contract)
python $REPO/tools/pmc_traffic.py $FCSV $WCSV $OUT/pmc_traffic_fc.json $OUT/pmc_hbm_traffic_fc.md \
    '{"strips": 1024, "ticks_per_step": 2048, "sample_rate": 48000, "fused": true, "eq_fast": false, "fp_contract": true, "n_gpus": 1, "gates_toggle": true}' audio
# 3c. what FETCH_SIZE says on KNOWN byte counts in the EqThree kernel's read pattern (half lines vs whole lines; tools/fetch_probe.hip)
( cd $REPO && bash tools/fetch_probe.sh 250 > /dev/null 2>&1 ); cp $REPO/gpurun_out/fetch_probe/times.txt $OUT/fetch_probe_times.txt; cp $REPO/gpurun_out/fetch_probe/counters.txt $OUT/fetch_probe_counters.txt
# 4. SQ counters of the headline kernels, gates toggling and held, and in the contracted order
SQCSV=$(pmc pmc_sq_toggle $SQ1 -- python $REPO/bench.py $COMMON); python $REPO/tools/pmc_sq_json.py $SQCSV $OUT/pmc_sq_toggle.json audio
pmc pmc_sq_held $SQ1 -- python $REPO/bench.py $COMMON --hold-gates > /dev/null
SQCSV=$(pmc pmc_sq_fc $SQ1 -- python $REPO/bench.py $COMMON --fp-contract); python $REPO/tools/pmc_sq_json.py $SQCSV $OUT/pmc_sq_fc.json audio
# 5. the clock the chip sustains under the headline kernels: GRBM_GUI_ACTIVE (per XCD) / duration of the same dispatches
CCSV=$(pmc pmc_clock GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -- python $REPO/bench.py $COMMON)
cp $(find /tmp/p_pmc_clock -name "*kernel_trace.csv" | head -1) $OUT/pmc_clock_kernel_trace.csv 2>/dev/null
python $REPO/tools/pmc_clock.py $CCSV $OUT/pmc_clock_kernel_trace.csv $OUT/clock.json
CCSV=$(pmc pmc_clock_fc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -- python $REPO/bench.py $COMMON --fp-contract)
python $REPO/tools/pmc_clock.py $CCSV $(find /tmp/p_pmc_clock_fc -name "*kernel_trace.csv" | head -1) $OUT/clock_fc.json
# 6. config 4 (video leg alone): kernel times, SQ counters, traffic
rm -rf /tmp/vk; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vk -- python $REPO/tools/vleg.py 4096 > $OUT/video_leg_line.json 2>/dev/null
cp $(find /tmp/vk -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_video_leg.csv
pmc video_sq $SQ1 -- python $REPO/tools/vleg.py 1280 > /dev/null
VF=$(pmc video_fetch FETCH_SIZE -- python $REPO/tools/vleg.py 1280)
VW=$(pmc video_write WRITE_SIZE -- python $REPO/tools/vleg.py 1280)
python $REPO/tools/pmc_traffic.py $VF $VW $OUT/video_pmc_traffic.json $OUT/video_pmc_hbm_traffic.md '{"leg": "video", "frames_per_launch": 16}' video
# 7. config 3 (FIR + resampler leg alone)
rm -rf /tmp/fk; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fk -- python $REPO/tools/fleg.py 128 10 > $OUT/fir_leg_line.json 2>/dev/null
cp $(find /tmp/fk -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_fir_leg.csv
pmc fir_sq $SQ1 -- python $REPO/tools/fleg.py 128 6 > /dev/null
pmc fir_sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -- python $REPO/tools/fleg.py 128 6 > /dev/null
# 7b. the resampler's limiter as counters (VERDICT r4 task 5): LDS conflicts / LDS array cycles / LDS issue stalls / LDS instructions, beside wave and busy cycles
pmc fir_sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- python $REPO/tools/fleg.py 128 6 > /dev/null
FF=$(pmc fir_fetch FETCH_SIZE -- python $REPO/tools/fleg.py 128 6)
FW=$(pmc fir_write WRITE_SIZE -- python $REPO/tools/fleg.py 128 6)
python $REPO/tools/pmc_traffic.py $FF $FW $OUT/fir_pmc_traffic.json $OUT/fir_pmc_hbm_traffic.md '{"leg": "fir_resample", "ticks_per_step": 128}' fir
# 7b. graph shapes beside the headline's (an Amplifier modulated by a buffer, launch groups of mixed epilogue modes, Envelopes gated by a module's output) and the
#     headline job at 44.1 kHz under the counters
( cd $REPO && timeout 600 python tools/ctl_probe.py 2048 > $OUT/graph_shapes_probe.txt 2>&1 )
( cd $REPO && timeout 900 bash tools/pmc_44k1.sh > $OUT/pmc_44k1.txt 2>&1 )
cd /tmp
# 8. LAST: the default command, as the driver runs it -- with this round's counter summaries in place, so that the line's roofline.traffic / limiter /
#    sustained clock are the ones just collected on these kernel sources (bench.py copies them only while the recorded source hash matches)
mkdir -p $REPO/profiles/$R && cp $OUT/*.json $REPO/profiles/$R/ 2>/dev/null
timeout 900 python $REPO/bench.py --full-out $OUT/bench_full.json > $OUT/bench_default_line.json 2> $OUT/bench_default.err
wc -c $OUT/bench_default_line.json
python $REPO/tools/profile_readme.py $OUT $R > $OUT/README.md
ls -la $OUT
