#!/bin/bash
# contracted / exact order at the headline shape: tile form (MX_EQ_SPEC_SB) x chunks per strip, gates toggling
cd $GRAFT_REPO_ROOT
for sb in 16 321; do
  MX_EQ_SPEC_SB=$sb python tools/eq_sweep.py --toggle --fp-contract --chunks 128,192,256 2>&1 | grep strips
done
for sb in 32 321; do
  MX_EQ_SPEC_SB=$sb python tools/eq_sweep.py --toggle --chunks 128,192,256 2>&1 | grep strips
done
