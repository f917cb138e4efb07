# kernel trace + SQ counters of one bench configuration: per-kernel durations of the steady state.  gpurun -- 'bash tools/ktrace.sh "--strips 128 --ticks-per-step 2048" [tag]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${2:-kt}; OUT=$R/gpurun_out/ktrace; mkdir -p $OUT
FL="$1 --steps 12 --warmup 3 --no-cpu-baseline --no-realtime --no-t-sweep --no-north-star --no-held-leg --no-material-leg --no-rate-leg --no-contract-leg --no-scaling-probe --fir-ticks 0 --repeats 0 --video-frames 0 --no-headline-parity"
python $R/bench.py $FL | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', round(l['value']/1e6,1), 'M  ms/step', round(l['ms_per_step'],4), l['roofline'].get('kernel_ms_per_step'))"
rm -rf /tmp/kt_$TAG; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python $R/bench.py $FL > /dev/null 2>&1
python - $(find /tmp/kt_$TAG -name "*kernel_trace.csv" | head -1) <<'PY' | tee $OUT/$TAG.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # the timed steps
seq = [(r["Kernel_Name"].split("(")[0][-46:], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")), r.get("LDS_Block_Size", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rows]
idx = [i for i, s in enumerate(seq) if "k_env_ticks" in s[0]]
if len(idx) > 3:
    a, b = idx[-3], idx[-2]
    t0 = seq[a][1]
    for n, s, e, v, l, g in seq[a:b + 1]:
        print(f"{n:48s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  vgpr {v} lds {l} grid {g}")
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e, *_ in seq: agg[n][0] += e - s; agg[n][1] += 1
for n, (t, c) in agg.items(): print(f"{n:48s} n {c:4d} mean {t / c / 1e3:7.1f} us")
PY
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"
rm -rf /tmp/pm_$TAG; timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/pm_$TAG -- python $R/bench.py $FL > /dev/null 2>&1
f=$(find /tmp/pm_$TAG -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep -A9 "spec_tiled" | tee -a $OUT/$TAG.txt
