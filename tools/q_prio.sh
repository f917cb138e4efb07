#!/bin/bash
# Mixer bank beside the next EqThree group (MX_FLAG_OVERLAP_TAIL), wall clock only (no per-group events), with the tail stream at low priority and / or the EqThree waves at s_setprio 3
cd $GRAFT_REPO_ROOT
run() { python tools/eq_sweep.py --toggle --steps 20 --no-profile "$@" 2>&1 | grep strips | sed 's/fast=False //; s/chunks=auto //; s/| spec.*=>/=>/'; }
for shape in "--ticks 2048" "--ticks 1024" "--ticks 256" "--ticks 128" "--ticks 2048 --strips 128" "--ticks 2048 --strips 256"; do
  echo "== $shape"
  run $shape
  run $shape --overlap-tail
  MX_TAIL_PRIO=1 run $shape --overlap-tail | sed 's/^/tailprio /'
  MX_EQ_PRIO=1 run $shape --overlap-tail | sed 's/^/eqprio /'
  MX_TAIL_PRIO=1 MX_EQ_PRIO=1 run $shape --overlap-tail | sed 's/^/both /'
done
