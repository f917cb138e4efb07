#!/bin/bash
# Mixer bank of run k beside run k + 1's EqThree group (MX_FLAG_OVERLAP_TAIL; held behind the gate the EqThree launch's last workgroup opens) against one stream; wall clock
cd $GRAFT_REPO_ROOT
run() { python tools/eq_sweep.py --toggle --steps 20 --no-profile "$@" 2>&1 | grep strips | sed 's/fast=False //; s/chunks=auto //; s/toggle=True //; s/sb=auto//; s/| spec.*=>/=>/'; }
for shape in "--ticks 2048" "--ticks 1024" "--ticks 512" "--ticks 256" "--ticks 128" "--ticks 64" "--ticks 32" "--ticks 2048 --strips 128" "--ticks 2048 --strips 256" "--ticks 2048 --strips 512" "--ticks 2048 --fp-contract" "--ticks 256 --fp-contract"; do
  run $shape
  run $shape --overlap-tail
done
echo "with per-group events (bench.py's timed region has them):"
python tools/eq_sweep.py --toggle --steps 20 --overlap-tail 2>&1 | grep strips
python tools/eq_sweep.py --toggle --steps 20 2>&1 | grep strips
