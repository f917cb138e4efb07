#!/bin/bash
# FETCH_SIZE on known byte counts in the speculative EqThree kernel's read pattern (tools/fetch_probe.hip):  gpurun -- 'bash tools/fetch_probe.sh [spin]'
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/fetch_probe; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_probe $REPO/tools/fetch_probe.hip 2>/dev/null || exit 1
SPIN=${1:-250}
/tmp/fetch_probe $SPIN | tee $OUT/times.txt
for ctr in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $ctr | tr ' ' '_'); rm -rf /tmp/fp_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/fp_$tag -- /tmp/fetch_probe $SPIN > /dev/null 2>&1
  f=$(find /tmp/fp_$tag -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "$ctr: no output"; continue; }
  python - "$f" <<'PY' | tee -a $OUT/counters.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:40]), {})[r["Counter_Name"]] = by.get((int(r["Dispatch_Id"]), r["Kernel_Name"][:40]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for (d, k), c in sorted(by.items()):
    print(d, k, {n: round(v) for n, v in c.items()})
PY
done
