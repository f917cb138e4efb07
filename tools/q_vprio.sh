#!/bin/bash
# wave priority per tile kind (MX_VIDEO_PRIO: bits 0-1 chain tiles, bits 2-3 scaler tiles) x row order (MX_VIDEO_ORDER: 0 chains first, 1 interleaved) in k_video_batch
cd $GRAFT_REPO_ROOT
for v in main no_rest_fader alpha; do
for cfg in "0 0" "1 4" "1 8" "1 12" "2 4"; do set -- $cfg; echo -n "$v order=$1 prio=$2: "; MX_VIDEO_ORDER=$1 MX_VIDEO_PRIO=$2 python tools/vleg.py 1920 3 $v 2>/dev/null | python -c "import sys,json; print([json.loads(l)['device_us_per_frame'] for l in sys.stdin if l.startswith('{')])"; done; done
