#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --force-combine --no-cpu-baseline --no-realtime --no-north-star --no-material-leg --no-rate-leg --no-contract-leg --fir-ticks 0 --repeats 2 --video-frames 0 --no-t-sweep --no-held-leg --no-headline-parity"
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],3), 'ms', d['repeats']['ms_per_step'] if d.get('repeats') else '', d['config']['overlap'][:30], d['exchange']['exchange_ms_per_step'])"; }
$B 2>/dev/null | show auto
MX_OVERLAP_AUTO=0 $B 2>/dev/null | show one-stream
$B --strips 128 2>/dev/null | show auto-128
MX_OVERLAP_AUTO=0 $B --strips 128 2>/dev/null | show one-stream-128
